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
    if t.dtype is torch.float32 and t.is_contiguous():     # (the common case: `.to` alone is ~9 us of dispatch, ~65 calls per step)
        return t.detach() if t.requires_grad else t
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


class SourceLoss(torch.autograd.Function):
    """(total [], terms [8]) = the whole loss of one source frame (scenerf.py:203-238 around process_single_source :243-320) from the
    renderer's outputs, one HIP launch each way (``scenerf_hip_source_loss_forward / _backward``); differentiable w.r.t. ``color``,
    ``depth``, ``loss_kl`` and ``gaussian_means``.  ``terms`` (detached) = {total, loss_reprojection, mean loss_color, mean loss_kl, mean
    dist2closest, mean min_som_vars, mean min_stds, n_valid}: what the trainer logs."""

    @staticmethod
    def forward(ctx, color, depth, loss_kl, gmeans, gstds, som_vars, pix_source, img_source, img_target, cam_K, inv_K, T_source2target, noise,
                noise_scale, w_rep, w_col, w_d2c, rng_state=None):
        for name, t in (("color", color), ("depth", depth), ("loss_kl", loss_kl), ("gaussian_means", gmeans), ("pix_source", pix_source),
                        ("img_source", img_source), ("img_target", img_target)):
            if not t.is_cuda:
                raise RuntimeError("%s must live on the GPU: the fused source-loss kernel has no CPU path" % name)
        lib = _capi.load()
        dev = color.device
        R = int(depth.numel())
        _, H, W = img_source.shape
        if tuple(img_target.shape) != tuple(img_source.shape) or img_source.shape[0] != 3:
            raise RuntimeError("img_source / img_target must both be (3, H, W)")
        col, dep, pix, kl, gm = _f32(color), _f32(depth).reshape(-1), _f32(pix_source), _f32(loss_kl).reshape(-1), _f32(gmeans)
        G = int(gm.shape[1])
        gs = _f32(gstds) if gstds is not None else None
        sv = _f32(som_vars) if som_vars is not None else None
        ims, imt = _f32(img_source), _f32(img_target)
        K, iK, T = _f32(cam_K), _f32(inv_K), _f32(T_source2target)
        nz = _f32(noise).reshape(-1) if noise is not None else None
        f = dict(dtype=torch.float32, device=dev)
        valid, dterm, col_src = torch.empty(R, **f), torch.empty(R, **f), torch.empty((R, 3), **f)
        closest = torch.empty(R, dtype=torch.int32, device=dev)
        partial, out8, total = torch.empty(8 * ((R + 63) // 64), **f), torch.empty(8, **f), torch.empty((), **f)
        if rng_state is not None and (rng_state.dtype != torch.int64 or rng_state.numel() != 2 or not rng_state.is_cuda):
            raise RuntimeError("rng_state must be a CUDA int64 tensor of two elements {seed, calls so far}")
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            _capi.check(lib.scenerf_hip_source_loss_forward(pix.data_ptr(), col.data_ptr(), dep.data_ptr(), kl.data_ptr(), gm.data_ptr(),
                                                            _capi.ptr(gs), _capi.ptr(sv), G, ims.data_ptr(), imt.data_ptr(), _capi.ptr(nz),
                                                            _capi.ptr(rng_state), float(noise_scale), K.data_ptr(), iK.data_ptr(), T.data_ptr(), R, H, W,
                                                            float(w_rep), float(w_col), float(w_d2c), valid.data_ptr(), dterm.data_ptr(),
                                                            col_src.data_ptr(), closest.data_ptr(), partial.data_ptr(), out8.data_ptr(),
                                                            total.data_ptr(), st), "source_loss_forward")
        ctx.save_for_backward(col, col_src, valid, dterm, gm, dep, closest, out8)
        ctx.set_materialize_grads(False)     # (no zero tensor -- a fill launch -- for the logged terms nobody differentiates)
        ctx.w = (float(w_rep), float(w_col), float(w_d2c))
        ctx.mark_non_differentiable(out8)
        return total, out8

    @staticmethod
    def backward(ctx, g_total, _g_terms):
        col, col_src, valid, dterm, gm, dep, closest, out8 = ctx.saved_tensors
        lib = _capi.load()
        R, G = int(valid.numel()), int(gm.shape[1])
        g_color, g_depth, g_kl, g_gm = torch.empty_like(col), torch.empty_like(valid), torch.empty_like(valid), torch.empty_like(gm)
        if g_total is None:     # only the (non-differentiable) terms received a gradient: nothing flows
            return (None,) * 18
        gt = _f32(g_total).reshape(1)
        with torch.cuda.device(col.device):
            st = torch.cuda.current_stream(col.device).cuda_stream
            _capi.check(lib.scenerf_hip_source_loss_backward(col.data_ptr(), col_src.data_ptr(), valid.data_ptr(), dterm.data_ptr(), gm.data_ptr(),
                                                             dep.data_ptr(), closest.data_ptr(), out8.data_ptr(), _capi.ptr(gt), R, G, ctx.w[0],
                                                             ctx.w[1], ctx.w[2], g_color.data_ptr(), g_depth.data_ptr(), g_kl.data_ptr(),
                                                             g_gm.data_ptr(), st), "source_loss_backward")
        return (g_color, g_depth, g_kl, g_gm) + (None,) * 14


def source_loss(out, pix_source: torch.Tensor, img_source: torch.Tensor, img_target: torch.Tensor, cam_K: torch.Tensor, inv_K: torch.Tensor,
                T_source2target: torch.Tensor, noise: Optional[torch.Tensor] = None, noise_scale: float = 1e-5, reproj_weight: float = 1.0,
                color_weight: float = 1.0, dist2closest_weight: float = 0.01,
                rng_state: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """The loss of one source frame from ``out`` = the dict of ``render_rays_batch`` (keys depth, color, loss_kl, gaussian_means,
    gaussian_stds, som_vars): (total, terms[8]).  ``noise`` [R] ~ N(0, 1) (scaled by ``noise_scale`` in the kernel: the reference adds
    ``randn * 1e-5`` to the identity term); the weights are those of the reference's ``forward`` (KITTI: 1, 1, 0.01; BundleFusion: 5, 1,
    0.1; a term switched off by ``use_reprojection`` / ``use_color`` = weight 0).  ``rng_state`` (with ``noise=None``): a CUDA int64 tensor
    ``[seed, calls so far]`` -- the noise is then made inside the kernel (Philox + Box-Muller) and the counter advanced by the launch:
    no ``randn`` launches in front of the kernel (``make_rng_state``)."""
    return SourceLoss.apply(out["color"], out["depth"].reshape(-1), out["loss_kl"], out["gaussian_means"], out.get("gaussian_stds"),
                            out.get("som_vars"), pix_source, img_source, img_target, cam_K, inv_K, T_source2target, noise, noise_scale,
                            reproj_weight, color_weight, dist2closest_weight, rng_state)


def make_rng_state(device, seed: Optional[int] = None) -> torch.Tensor:
    """``[seed, 0]`` for ``source_loss(rng_state=...)``; the seed is taken from torch's CPU generator (so ``torch.manual_seed`` makes runs
    repeatable) unless given."""
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    return torch.tensor([int(seed), 0], dtype=torch.int64, device=device)
