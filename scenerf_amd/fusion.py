"""TSDF fusion on the MI355X: drop-in for ``scenerf.data.utils.fusion.TSDFVolume`` (reference
scenerf/data/utils/fusion.py:20-391) as the reconstruction scripts use it (scripts/reconstruction/depth2tsdf*.py,
generate_sc_gt_bf.py): ``TSDFVolume(vol_bnds, voxel_size, trunc_margin=10)``, ``integrate(color_im, depth_im, cam_intr, cam_pose,
obs_weight=1.)``, ``get_volume()``.  The volumes live in HBM for the whole sequence; a frame costs two small H2D copies and one
kernel (the reference re-uploads every argument array with ``cuda.InOut`` on every call, fusion.py:223-235).

``semantics="gpu"`` (default) follows the reference's pycuda kernel, which is what runs on a CUDA machine; ``"cpu"`` follows the
CPU path of the same file (a different update rule, see csrc/tsdf.hip).  ``get_point_cloud`` / ``get_mesh`` need scikit-image's
marching cubes exactly like the reference and are forwarded to it when it is installed.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _capi

COLOR_CONST = 256 * 256


class TSDFVolume:
    def __init__(self, vol_bnds, voxel_size, trunc_margin=10, use_gpu=True, semantics="gpu", device="cuda"):
        vol_bnds = np.asarray(vol_bnds, dtype=np.float64)
        assert vol_bnds.shape == (3, 2), "[!] `vol_bnds` should be of shape (3, 2)."
        if not use_gpu:
            raise RuntimeError("scenerf_amd.fusion.TSDFVolume only runs on the GPU (no CPU fallback in the product path)")
        if semantics not in ("gpu", "cpu"):
            raise ValueError("semantics must be 'gpu' (the reference's pycuda kernel) or 'cpu' (its CPU path)")
        self._semantics = 0 if semantics == "gpu" else 1
        self._vol_bnds = vol_bnds.copy()
        self._voxel_size = float(voxel_size)
        self._trunc_margin = float(trunc_margin)
        self._color_const = COLOR_CONST
        # fusion.py:43-46
        self._vol_dim = np.ceil((self._vol_bnds[:, 1] - self._vol_bnds[:, 0]) / self._voxel_size).copy(order="C").astype(int)
        self._vol_bnds[:, 1] = self._vol_bnds[:, 0] + self._vol_dim * self._voxel_size
        self._vol_origin = self._vol_bnds[:, 0].copy(order="C").astype(np.float32)
        self.device = torch.device(device)
        dims = tuple(int(d) for d in self._vol_dim)
        self._tsdf = torch.full(dims, 255.0, dtype=torch.float32, device=self.device)     # fusion.py:53
        self._weight = torch.zeros(dims, dtype=torch.float32, device=self.device)
        self._color = torch.zeros(dims, dtype=torch.float32, device=self.device)
        self._lib = _capi.load()
        self.gpu_mode = 1

    def integrate(self, color_im, depth_im, cam_intr, cam_pose, obs_weight=1.):
        """fusion.py:205-325.  color_im (H, W, 3) RGB, depth_im (H, W) metres (0 = invalid), cam_intr (3, 3), cam_pose (4, 4)."""
        depth = torch.as_tensor(np.ascontiguousarray(depth_im, dtype=np.float32)).to(self.device, non_blocking=True)
        col = torch.as_tensor(np.ascontiguousarray(color_im)).to(self.device, non_blocking=True).to(torch.float32)
        folded = torch.floor(col[..., 2] * self._color_const + col[..., 1] * 256 + col[..., 0]).contiguous()   # fusion.py:219-220
        im_h, im_w = depth.shape
        dim = (C.c_int32 * 3)(*[int(d) for d in self._vol_dim])
        org = (C.c_float * 3)(*[float(x) for x in self._vol_origin])
        K = (C.c_float * 9)(*[float(x) for x in np.asarray(cam_intr, dtype=np.float32).reshape(-1)])
        P = (C.c_float * 16)(*[float(x) for x in np.asarray(cam_pose, dtype=np.float32).reshape(-1)])
        Pi = (C.c_double * 16)(*[float(x) for x in np.linalg.inv(np.asarray(cam_pose)).reshape(-1)])          # fusion.py:239
        _capi.check(self._lib.scenerf_hip_tsdf_integrate(self._tsdf.data_ptr(), self._weight.data_ptr(), self._color.data_ptr(),
                                                        C.byref(dim), C.byref(org), self._voxel_size, C.byref(K), C.byref(P), C.byref(Pi),
                                                        folded.data_ptr(), depth.data_ptr(), int(im_h), int(im_w), self._trunc_margin,
                                                        float(obs_weight), self._semantics,
                                                        torch.cuda.current_stream(self.device).cuda_stream), "tsdf_integrate")

    def get_volume(self):
        """fusion.py:327-331: (tsdf, colour) as numpy arrays."""
        return self._tsdf.cpu().numpy(), self._color.cpu().numpy()

    def get_weight(self):
        return self._weight.cpu().numpy()

    def get_point_cloud(self):
        """fusion.py:333-356 (needs scikit-image, like the reference)."""
        from skimage import measure
        tsdf_vol, color_vol = self.get_volume()
        verts = measure.marching_cubes(tsdf_vol, level=0)[0]
        verts_ind = np.round(verts).astype(int)
        verts = verts * self._voxel_size + self._vol_origin
        rgb = color_vol[verts_ind[:, 0], verts_ind[:, 1], verts_ind[:, 2]]
        b = np.floor(rgb / self._color_const)
        g = np.floor((rgb - b * self._color_const) / 256)
        r = rgb - b * self._color_const - g * 256
        return verts, np.floor(np.asarray([r, g, b])).T.astype(np.uint8)
