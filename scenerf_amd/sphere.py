"""Image -> sphere resampling of the encoder levels on the MI355X: drop-in for
``DecoderSphere.get_sphere_feature(x, pix, pix_sphere, scale)`` (reference scenerf/models/unet2d_sphere.py:138-165), which the
decoder calls six times per image (levels 1 ... 32).

The reference re-scatters the 451 k pixel coordinates into a fresh map on every call; the map only depends on the camera
intrinsics, so it is built once per (pix, pix_sphere, level, plane size) by ``scenerf_hip_sphere_map_build`` and cached -- a call is
then ONE gather kernel forward and ONE gather kernel backward (no atomics, deterministic).  Duplicate cells resolve to the last
pixel in index order, the reference's single-thread CPU behaviour (its CUDA scatter has no defined winner).  fp32 only, like the
decoder that consumes it; there is no CPU / eager fallback.

    rs = SphereResampler(out_img_W=1500, out_img_H=452)
    feats = rs.get_sphere_feature(x, pix, pix_sphere, scale)      # (B, C, round(452/scale), round(1500/scale)), differentiable in x
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import _capi


class SphereMap:
    """The cached scatter result of one level: ``src`` (out_h, out_w) int32 = (sy << 16 | sx) or -1, plus (lazily) the cells
    grouped by source pixel for the backward gather.  The CSR is indexed on a (H+1) x (W+1) grid so that a map pointing one past
    the plane (whose in-range taps still count, as in grid_sample's zero padding) needs no special case."""

    def __init__(self, src: torch.Tensor, H: int, W: int):
        self.src, self.H, self.W = src, H, W
        self.out_h, self.out_w = src.shape
        self._csr: Optional[Tuple[torch.Tensor, torch.Tensor]] = None

    def csr(self) -> Tuple[torch.Tensor, torch.Tensor]:
        if self._csr is None:
            flat = self.src.reshape(-1)
            cells = torch.nonzero(flat >= 0).squeeze(1)
            sy, sx = (flat[cells] >> 16).long(), (flat[cells] & 0xFFFF).long()
            keep = (sy <= self.H) & (sx <= self.W)          # further out: no tap inside the plane
            cells, q = cells[keep], (sy * (self.W + 1) + sx)[keep]
            order = torch.argsort(q, stable=True)           # ascending cell index inside a group: fixed summation order
            n = (self.H + 1) * (self.W + 1)
            row_ptr = torch.zeros(n + 1, dtype=torch.int32, device=flat.device)
            row_ptr[1:] = torch.cumsum(torch.bincount(q, minlength=n), 0).to(torch.int32)
            self._csr = (row_ptr, cells[order].to(torch.int32).contiguous())
        return self._csr


def scaled_dims(out_img_W: int, out_img_H: int, scale: int) -> Tuple[int, int]:
    """unet2d_sphere.py:139 (Python round, half to even: 1500 x 452 at level 8 is 188 x 56)."""
    return round(out_img_W / scale), round(out_img_H / scale)


def build_map(pix: torch.Tensor, pix_sphere: torch.Tensor, scale: int, out_w: int, out_h: int, H: int, W: int) -> SphereMap:
    """unet2d_sphere.py:140-147 on the device."""
    if not pix.is_cuda:
        raise RuntimeError("scenerf_amd.sphere runs on the GPU only (no CPU fallback in the product path)")
    lib = _capi.load()
    pix = pix.to(torch.float32).contiguous()
    ps = pix_sphere.to(device=pix.device, dtype=torch.int64).contiguous()
    assert pix.shape == ps.shape and pix.dim() == 2 and pix.shape[1] == 2, "pix / pix_sphere must be (P, 2)"
    winner = torch.empty(out_h * out_w, dtype=torch.int32, device=pix.device)
    src = torch.empty((out_h, out_w), dtype=torch.int32, device=pix.device)
    _capi.check(lib.scenerf_hip_sphere_map_build(pix.data_ptr(), ps.data_ptr(), pix.shape[0], int(scale), out_w, out_h,
                                                 winner.data_ptr(), src.data_ptr(), torch.cuda.current_stream(pix.device).cuda_stream),
                "sphere_map_build")
    return SphereMap(src, H, W)


class _Resample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: torch.Tensor, m: SphereMap, hwc: bool = False) -> torch.Tensor:
        if not x.is_cuda or x.dtype != torch.float32:
            raise RuntimeError("sphere resampling needs a float32 CUDA tensor (got %s on %s); no CPU / eager fallback" % (x.dtype, x.device))
        B, C, H, W = x.shape
        assert (H, W) == (m.H, m.W), "map was built for a %dx%d plane, got %dx%d" % (m.H, m.W, H, W)
        x = x.contiguous()
        lib, st = _capi.load(), torch.cuda.current_stream(x.device).cuda_stream
        if hwc:   # channels-last sphere side: what the renderer reads in place (renderer.HWC)
            out = torch.empty((B, m.out_h, m.out_w, C), dtype=torch.float32, device=x.device)
            _capi.check(lib.scenerf_hip_sphere_resample_forward_nhwc(x.data_ptr(), B * C, C, H, W, m.src.data_ptr(), m.out_w, m.out_h,
                                                                     out.data_ptr(), st), "sphere_resample_forward_nhwc")
        else:
            out = torch.empty((B, C, m.out_h, m.out_w), dtype=torch.float32, device=x.device)
            _capi.check(lib.scenerf_hip_sphere_resample_forward(x.data_ptr(), B * C, H, W, m.src.data_ptr(), m.out_w, m.out_h,
                                                                out.data_ptr(), st), "sphere_resample_forward")
        ctx.m, ctx.shape, ctx.hwc = m, (B, C, H, W), hwc
        return out

    @staticmethod
    def backward(ctx, dout: torch.Tensor):
        m, (B, C, H, W) = ctx.m, ctx.shape
        row_ptr, cells = m.csr()
        dout = dout.contiguous().float()
        dx = torch.empty((B, C, H, W), dtype=torch.float32, device=dout.device)
        lib, st = _capi.load(), torch.cuda.current_stream(dout.device).cuda_stream
        if ctx.hwc:
            _capi.check(lib.scenerf_hip_sphere_resample_backward_nhwc(dout.data_ptr(), B * C, C, H, W, row_ptr.data_ptr(), cells.data_ptr(),
                                                                      m.out_w, m.out_h, dx.data_ptr(), st), "sphere_resample_backward_nhwc")
        else:
            _capi.check(lib.scenerf_hip_sphere_resample_backward(dout.data_ptr(), B * C, H, W, row_ptr.data_ptr(), cells.data_ptr(),
                                                                 m.out_w, m.out_h, dx.data_ptr(), st), "sphere_resample_backward")
        return dx, None, None


def resample(x: torch.Tensor, m: SphereMap, hwc: bool = False) -> torch.Tensor:
    return _Resample.apply(x, m, hwc)


class SphereResampler:
    """Holds the per-level maps of one camera.  ``get_sphere_feature`` has the reference method's signature; the cache is keyed on
    the identity and version of ``pix`` / ``pix_sphere`` (the encoder grid of ``SphericalMapping.from_pixels``), the level and the
    plane size, with the few most recent geometries kept."""

    def __init__(self, out_img_W: int, out_img_H: int, max_cached: int = 16, layout: str = "chw"):
        if layout not in ("chw", "hwc"):
            raise ValueError("layout must be 'chw' (the reference's (B, C, h, w)) or 'hwc' ((B, h, w, C), for renderer.HWC)")
        self.layout = layout
        self.out_img_W, self.out_img_H, self.max_cached = out_img_W, out_img_H, max_cached
        self._maps: Dict[tuple, tuple] = {}

    def map_for(self, pix: torch.Tensor, pix_sphere: torch.Tensor, scale: int, H: int, W: int) -> SphereMap:
        key = (id(pix), pix._version, id(pix_sphere), pix_sphere._version, int(scale), H, W)
        hit = self._maps.get(key)
        if hit is not None and hit[1] is pix and hit[2] is pix_sphere:
            return hit[0]
        out_w, out_h = scaled_dims(self.out_img_W, self.out_img_H, scale)
        m = build_map(pix, pix_sphere, scale, out_w, out_h, H, W)
        if len(self._maps) >= self.max_cached:
            self._maps.pop(next(iter(self._maps)))
        self._maps[key] = (m, pix, pix_sphere)          # the tensors are kept alive so their ids cannot be reused
        return m

    def get_sphere_feature(self, x: torch.Tensor, pix: torch.Tensor, pix_sphere: torch.Tensor, scale: int) -> torch.Tensor:
        """unet2d_sphere.py:138-165.  x (B, C, H, W) float32 -> (B, C, round(out_img_H/scale), round(out_img_W/scale)); with
        ``layout="hwc"`` the same values as (B, h, w, C): ``x_rgb["1_%d" % scale] = HWC(out[i])`` hands level ``scale`` of batch item ``i``
        to ``render_rays_batch`` without any layout conversion."""
        return resample(x, self.map_for(pix, pix_sphere, scale, x.shape[2], x.shape[3]), self.layout == "hwc")
