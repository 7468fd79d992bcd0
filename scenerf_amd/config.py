"""Hot-path constants of SceneRF (what the reference keeps as LightningModule attributes).

Mirrors the ctor arguments of reference scenerf/models/scenerf.py:23-116 (KITTI) and
scenerf/models/scenerf_bf.py (BundleFusion) that reach ``render_rays_batch``.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Tuple

from . import _capi

FEAT_SCALES = (1, 2, 4, 8, 16)
FEAT_CHANNELS = (80, 160, 320, 640, 1280)

# SphericalMapping FOV constants hard-coded in the reference (scenerf.py:83-88, scenerf_bf.py:85-90)
KITTI_FOV = dict(v_angle_max=104.7294, v_angle_min=75.4815, h_angle_max=131.1128, h_angle_min=49.5950)
BF_FOV = dict(v_angle_max=112.2911, v_angle_min=67.6248, h_angle_max=118.6861, h_angle_min=61.2383)


@dataclass
class RenderConfig:
    img_size: Tuple[int, int] = (1220, 370)
    sphere_W: int = 1500
    sphere_H: int = 452
    v_angle_max: float = KITTI_FOV["v_angle_max"]
    v_angle_min: float = KITTI_FOV["v_angle_min"]
    h_angle_max: float = KITTI_FOV["h_angle_max"]
    h_angle_min: float = KITTI_FOV["h_angle_min"]
    add_fov_hor: float = 0.0
    add_fov_ver: float = 0.0
    n_pts_uni: int = 32
    n_gaussians: int = 4
    n_pts_per_gaussian: int = 8
    max_sample_depth: float = 100.0
    std: float = 2.5
    som_sigma: float = 2.0
    gauss_floor: float = 1.5     # scenerf.py:591-594; 0.5 in scenerf_bf.py:606-608
    # uniform samples drawn when n_pts_uni == 0: the BundleFusion model substitutes 2 (scenerf_bf.py:623-626); the KITTI model has no
    # substitute and divides by zero there (utils.py:77)
    uni_fallback: int = 0
    kl_std_floor: float = 1.5    # ray_som_kl.py:83
    precision: str = "bf16"      # "bf16": bf16 GEMM operands / fp32 accumulate; "fp32": fp32 MFMA everywhere
    device_rng: bool = False     # False: gaussian noise drawn on CPU like the reference (utils.py:208-211)
    # pyramid levels (indices into FEAT_SCALES) that are NOT converted to (H,W,C): the gather reads the caller's fp32 (C,H,W) map
    # and the feature gradient lands in a (C,H,W) buffer directly.  Quirk Q1 (full-resolution sample index against a down-scaled
    # divisor, scenerf.py:522-525) keeps scale 1/s in range for at most 1/s^2 of the sphere (KITTI geometry: 0.14 % of the samples
    # at 1/4, none at 1/8 and 1/16), so for the two coarsest levels the per-image conversion, the accumulator transpose and the
    # half-precision copy cost more than the (rare to absent) strided accesses.  Measured at KITTI: (3, 4) saves 45 us per step;
    # adding level 2 saves 80 us more in conversions but its 0.14 % of samples cluster in a few row tiles whose 1280 strided
    # cache-line reads per sample stretch the gather by 100 us.
    direct_scales: Tuple[int, ...] = (3, 4)
    # pyramid levels handed over as fp32 (H,W,C) tensors (renderer.HWC): read in place, gradient returned in (H,W,C) -- no layout
    # conversion in either direction (scenerf_cfg.map_chw == 2).  Set per call by RenderSession from the maps it is given.
    hwc_scales: Tuple[int, ...] = ()
    # kernel-path selection (scenerf_cfg.fused_min_rows / fwd_kernel / flags): explicit per-call state, no environment variables
    fused_min_rows: int = _capi.FUSED_MIN_ROWS_DEFAULT   # bf16: rows from which the ResnetFC trunk / dgrad chain run as one fused kernel; < 0 never
    fwd_kernel: str = "wide"          # fused forward variant: "ring" (fused.hip: 64-row blocks), "wide" (wide.hip: 128-row blocks)
    fused_backward: bool = True       # False: the dgrad chain as six per-layer GEMMs even where the fused chain applies
    wgrad_tr: bool = True             # False: weight gradients through gemm_tn only
    dfeat_gemm: bool = False          # True: bf16 feature-map gradients through gemm.hip's scatter epilogue instead of dfeat.hip
    dfeat_per_scale: bool = False     # True: feature-gradient GEMM + scatter as one launch per pyramid level
    wgrad_overlap: bool = False       # True: per-layer backward runs the weight-gradient GEMMs on an internal side stream
    wide_any_m: bool = False          # True: the 128-row kernels also for launches of fewer than 192 row blocks (tests, probes)
    # fused dgrad chain: "ring" (fused.hip, 64-row blocks), "wide" (wide.hip, 128-row blocks; lin_out's input gradient is made in the
    # kernel's prologue from d_logits and H3's sign bits) or "wide_staged" (the 128-row chain on a dH3 tile written by linout_bwd like the
    # ring's: tests and A/B runs).  ring and wide_staged are bit-identical to the per-layer dgrad GEMMs; wide differs from them only
    # where the prologue's fp32-accurate K = 4 product rounds to a different bf16 than linout_bwd's FMA chain
    bwd_kernel: str = "wide"

    # ---- derived -----------------------------------------------------------------------------------------
    @property
    def uniform_only(self) -> bool:
        """scenerf.py:647-650 / scenerf_bf.py:662-665: with n_pts_uni == 0 and n_pts_per_gaussian == 1 the reference renders the
        uniform samples alone (the gaussian head still runs: its means / stds feed the KL term)."""
        return self.n_pts_uni == 0 and self.n_pts_per_gaussian == 1

    @property
    def n_uni_drawn(self) -> int:
        """Uniform samples the reference draws per ray: n_pts_uni, or the variant's substitute when that is 0."""
        return self.n_pts_uni if self.n_pts_uni > 0 else self.uni_fallback

    @property
    def n_uni_used(self) -> int:
        """Uniform samples that reach the renderer (0 in the gaussian-only branch, where the substitute draw is discarded)."""
        return self.n_pts_uni if self.n_pts_uni > 0 else (self.uni_fallback if self.uniform_only else 0)

    @property
    def n_samples(self) -> int:
        if self.uniform_only:
            return self.n_uni_drawn
        u = self.n_pts_uni if self.n_pts_uni > 0 else 0
        return u + self.n_gaussians * self.n_pts_per_gaussian

    @property
    def fov(self):
        v_max = self.v_angle_max + self.add_fov_ver
        v_min = self.v_angle_min - self.add_fov_ver
        h_max = self.h_angle_max + self.add_fov_hor
        h_min = self.h_angle_min - self.add_fov_hor
        return v_min, abs(v_max - v_min), h_min, abs(h_max - h_min)

    @property
    def precision_code(self) -> int:
        if self.precision not in ("bf16", "fp32"):
            raise ValueError("precision must be 'bf16' or 'fp32', got %r" % (self.precision,))
        return 1 if self.precision == "bf16" else 0

    def map_shapes(self):
        """(C, H, W) of the 5 encoder maps: round(size/scale), unet2d_sphere.py:139."""
        return [(c, round(self.sphere_H / s), round(self.sphere_W / s)) for s, c in zip(FEAT_SCALES, FEAT_CHANNELS)]

    def validate(self) -> None:
        if not (1 <= self.n_gaussians <= 8):
            raise ValueError("n_gaussians must be in [1, 8]")
        if self.n_pts_per_gaussian < 1 or self.n_pts_uni < 0:
            raise ValueError("bad sample counts")
        if self.uniform_only and self.n_uni_drawn == 0:
            # the KITTI model: sample_rays_viewdir(n_pts_per_ray=0) divides by zero (utils.py:77, step = (d_max - d_min) / 0) before the
            # uniform-only branch is reached -- the same exception here
            raise ZeroDivisionError("n_pts_uni == 0 with n_pts_per_gaussian == 1 renders the uniform samples alone (scenerf.py:647-650) "
                                    "and this model draws none: float division by zero (utils.py:77)")
        if self.n_samples > 512:
            raise ValueError("n_samples = %d exceeds the 512-sample limit of the wave-per-ray kernels" % self.n_samples)
        _ = self.precision_code
        if self.fwd_kernel not in ("ring", "wide"):
            raise ValueError("fwd_kernel must be 'ring' or 'wide', got %r" % (self.fwd_kernel,))
        if self.bwd_kernel not in ("ring", "wide", "wide_staged"):
            raise ValueError("bwd_kernel must be 'ring', 'wide' or 'wide_staged', got %r" % (self.bwd_kernel,))

    def uses_fused(self, rows: int) -> bool:
        """Whether an MLP pass over ``rows`` rows runs on the fused kernels (mirrors srf_use_fused in csrc/common.h)."""
        th = _capi.FUSED_MIN_ROWS_DEFAULT if self.fused_min_rows == 0 else self.fused_min_rows
        return self.precision_code == 1 and th > 0 and rows >= th

    def to_c(self) -> "_capi.Cfg":
        self.validate()
        c = _capi.Cfg()
        c.n_pts_uni = self.n_uni_used
        c.n_gaussians = self.n_gaussians
        c.n_pts_per_gaussian = self.n_pts_per_gaussian
        c.n_samples = self.n_samples
        c.sphere_W, c.sphere_H = self.sphere_W, self.sphere_H
        c.max_sample_depth = self.max_sample_depth
        c.uni_step = (self.max_sample_depth - 0.2) / self.n_uni_used if self.n_uni_used > 0 else 0.0
        c.base_std = self.std
        c.som_sigma = self.som_sigma
        c.gauss_floor = self.gauss_floor
        c.kl_std_floor = self.kl_std_floor
        c.v_min, c.v_fov, c.h_min, c.h_fov = self.fov
        for i, (s, (ch, h, w)) in enumerate(zip(FEAT_SCALES, self.map_shapes())):
            c.map_C[i], c.map_H[i], c.map_W[i] = ch, h, w
            c.div_H[i], c.div_W[i] = self.sphere_H // s, self.sphere_W // s   # scenerf.py:525
        c.precision = self.precision_code
        for i in range(5):
            c.map_chw[i] = 2 if i in self.hwc_scales else (1 if i in self.direct_scales else 0)
        c.fused_min_rows = int(self.fused_min_rows)
        c.fwd_kernel = {"ring": 0, "wide": 2}[self.fwd_kernel]
        c.flags = ((0 if self.fused_backward else _capi.FLAG_NO_FUSED_BWD) | (0 if self.wgrad_tr else _capi.FLAG_NO_WGRAD_TR)
                   | (_capi.FLAG_DFEAT_PER_SCALE if self.dfeat_per_scale else 0) | (_capi.FLAG_WGRAD_OVERLAP if self.wgrad_overlap else 0)
                   | (_capi.FLAG_UNIFORM_ONLY if self.uniform_only else 0)
                   | (_capi.FLAG_WIDE_BWD if self.bwd_kernel in ("wide", "wide_staged") else 0)
                   | (_capi.FLAG_WIDE_BWD_STAGED if self.bwd_kernel == "wide_staged" else 0) | (_capi.FLAG_WIDE_ANY_M if self.wide_any_m else 0) | (_capi.FLAG_DFEAT_GEMM if self.dfeat_gemm else 0))
        return c

    @staticmethod
    def kitti(**kw) -> "RenderConfig":
        d = dict(add_fov_hor=20.0, add_fov_ver=8.0, std=2.0, som_sigma=2.0)  # scripts/train_kitti.py defaults
        d.update(kw)
        return RenderConfig(**d)

    @staticmethod
    def bundlefusion(**kw) -> "RenderConfig":
        d = dict(img_size=(640, 480), sphere_W=960, sphere_H=720, add_fov_hor=14.0, add_fov_ver=11.0,
                 max_sample_depth=12.0, std=0.1, som_sigma=0.02, gauss_floor=0.5, uni_fallback=2, **BF_FOV)
        d.update(kw)
        return RenderConfig(**d)
