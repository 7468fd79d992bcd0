"""Loss-side gathers of a training step fused with the renderer's per-ray outputs: HIP kernel behind an autograd
Function.  reference scenerf/models/scenerf.py:302-307 (colour L1 against ``sample_pix_features`` of the source image) and
``compute_reprojection_loss`` (scenerf.py:349-386, with ``cam_pts_2_pix`` utils.py:298-315 and ``sample_pix_features`` utils.py:250-266).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _capi


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


class LossSide(torch.autograd.Function):
    """(loss_color [R,3], loss_reprojection []) = f(color [R,3], depth [R]; pix_source, img_source, img_target, cam_K, inv_K,
    T_source2target, noise).  Differentiable w.r.t. ``color`` and ``depth`` (the images carry no gradient in the reference)."""

    @staticmethod
    def forward(ctx, color, depth, pix_source, img_source, img_target, cam_K, inv_K, T_source2target, noise):
        for name, t in (("color", color), ("depth", depth), ("pix_source", pix_source), ("img_source", img_source), ("img_target", img_target)):
            if not t.is_cuda:
                raise RuntimeError("%s must live on the GPU: the fused loss-side kernel has no CPU path" % name)
        lib = _capi.load()
        dev = color.device
        R = int(depth.numel())
        _, H, W = img_source.shape
        if tuple(img_target.shape) != tuple(img_source.shape) or img_source.shape[0] != 3:
            raise RuntimeError("img_source / img_target must both be (3, H, W)")
        col, dep, pix = _f32(color), _f32(depth).reshape(-1), _f32(pix_source)
        ims, imt = _f32(img_source), _f32(img_target)
        K, iK, T = _f32(cam_K), _f32(inv_K), _f32(T_source2target)
        nz = _f32(noise).reshape(-1) if noise is not None else None
        f = dict(dtype=torch.float32, device=dev)
        loss_color = torch.empty((R, 3), **f)
        ray_term, valid, dterm = torch.empty(R, **f), torch.empty(R, **f), torch.empty(R, **f)
        col_src = torch.empty((R, 3), **f)
        acc2, loss_rep = torch.empty(2, **f), torch.empty((), **f)
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            _capi.check(lib.scenerf_hip_loss_side_forward(pix.data_ptr(), col.data_ptr(), dep.data_ptr(), ims.data_ptr(), imt.data_ptr(),
                                                          nz.data_ptr() if nz is not None else None, K.data_ptr(), iK.data_ptr(), T.data_ptr(),
                                                          R, H, W, loss_color.data_ptr(), ray_term.data_ptr(), valid.data_ptr(),
                                                          dterm.data_ptr(), col_src.data_ptr(), acc2.data_ptr(), loss_rep.data_ptr(), st),
                        "loss_side_forward")
        ctx.save_for_backward(col, col_src, valid, dterm, acc2)
        ctx.col_src = col_src
        return loss_color, loss_rep

    @staticmethod
    def backward(ctx, g_lc, g_lr):
        col, col_src, valid, dterm, acc2 = ctx.saved_tensors
        lib = _capi.load()
        R = int(valid.numel())
        g_color, g_depth = torch.empty_like(col), torch.empty_like(valid)
        glc = _f32(g_lc) if g_lc is not None else None
        glr = _f32(g_lr).reshape(1) if g_lr is not None else None
        with torch.cuda.device(col.device):
            st = torch.cuda.current_stream(col.device).cuda_stream
            _capi.check(lib.scenerf_hip_loss_side_backward(col.data_ptr(), col_src.data_ptr(), valid.data_ptr(), dterm.data_ptr(), acc2.data_ptr(),
                                                           glc.data_ptr() if glc is not None else None,
                                                           glr.data_ptr() if glr is not None else None, R, g_color.data_ptr(),
                                                           g_depth.data_ptr(), st), "loss_side_backward")
        return g_color, g_depth, None, None, None, None, None, None, None


def loss_side(color: torch.Tensor, depth: torch.Tensor, pix_source: torch.Tensor, img_source: torch.Tensor, img_target: torch.Tensor,
              cam_K: torch.Tensor, inv_K: torch.Tensor, T_source2target: torch.Tensor,
              noise: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """(loss_color [R,3], loss_reprojection scalar) of scenerf.py:302-307 / 349-386 in one kernel.  ``noise`` [R]: the values the
    reference adds to the identity term (``torch.randn(R) * 1e-5``); None = no noise."""
    return LossSide.apply(color, depth.reshape(-1), pix_source, img_source, img_target, cam_K, inv_K, T_source2target, noise)
