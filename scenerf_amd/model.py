"""Drop-in boundary: a ``SceneRF`` module with the reference's constructor, attributes, state_dict keys and
``render_rays_batch`` signature (reference scenerf/models/scenerf.py:22-116, 392-471; BundleFusion variant
scenerf/models/scenerf_bf.py), whose hot path runs in libscenerf_hip.so.

What is kept identical for callers:
  * ctor kwargs and defaults; attributes ``spherical_mapping, net_rgb, mlp, mlp_gaussian, pe, ray_som, n_rays,
    img_size`` read by the evaluation / reconstruction scripts;
  * parameter / buffer names: ``mlp.lin_in``, ``mlp.lin_z.{0,1,2}``, ``mlp.blocks.{0,1,2}.fc_{0,1}``,
    ``mlp.lin_out`` (same under ``mlp_gaussian.``), ``pe._freqs``, ``pe._phases`` -> reference checkpoints load;
  * ``render_rays_batch(cam_K, T_source2infer, x_rgb, depth_window=100, T_cam2velo=None, sampled_pixels=None,
    ray_batch_size=128) -> dict`` with the 12 keys of scenerf.py:456-469.
The image encoder (``net_rgb``) and the losses stay stock PyTorch and are injected by the caller.

Two keyword arguments the reference does not have: ``precision`` -- "fp32" (default: fp32 MFMA end to end, the reference's own
numerics, SURVEY 8d fp32 gates) or "bf16" (bf16 MFMA operands / fp32 accumulate, the fused kernels: BASELINE.json configs[1], what
bench.py times; an explicit opt-in because it changes training numerics) -- and ``device_rng``.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from .config import BF_FOV, KITTI_FOV, RenderConfig
from .renderer import MLP_PARAM_NAMES, OUTPUT_KEYS, RenderSession
from .training import BundleFusionTrainingMixin, TrainingMixin

try:  # the reference derives from pl.LightningModule; Lightning is optional here
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # pragma: no cover - Lightning is not installed in the build image
    class _Base(nn.Module):
        """What the evaluation / reconstruction scripts and a plain training loop use of ``pl.LightningModule`` when Lightning is not
        installed: ``hparams`` / ``save_hyperparameters``, ``log`` (a no-op), ``device`` and ``load_from_checkpoint`` for
        Lightning-format checkpoints (``{"state_dict": ..., "hyper_parameters": ...}``)."""

        def save_hyperparameters(self, *a, ignore=(), **k):
            # Lightning's behaviour for a bare call: the arguments of every __init__ frame of this object, innermost last
            import inspect
            hp = {}
            fr = inspect.currentframe().f_back
            frames = []
            while fr is not None and fr.f_code.co_name == "__init__" and fr.f_locals.get("self") is self:
                frames.append(fr)
                fr = fr.f_back
            for fr in reversed(frames):     # outermost (the subclass) first, so that what it forwards does not shadow its own arguments
                info = inspect.getargvalues(fr)
                for name in info.args[1:]:
                    hp.setdefault(name, info.locals[name])
                if info.keywords:
                    for name, v in info.locals[info.keywords].items():
                        hp.setdefault(name, v)
            for name in ([ignore] if isinstance(ignore, str) else list(ignore)):
                hp.pop(name, None)
            self.hparams = hp

        def log(self, *a, **k):
            pass

        @property
        def device(self):
            return next(self.parameters()).device

        @classmethod
        def load_from_checkpoint(cls, checkpoint_path, map_location=None, hparams_file=None, strict=True, **kwargs):
            """``pl.LightningModule.load_from_checkpoint`` for the callers of the reference (render_colors.py:38-40,
            save_depth_metrics.py:57, generate_novel_depths.py:48): constructor arguments from the checkpoint's ``hyper_parameters``
            (overridden by ``kwargs``; arguments this constructor does not have are dropped), then ``load_state_dict``."""
            import inspect
            if hparams_file is not None:
                raise NotImplementedError("hparams_file is not supported without pytorch_lightning")
            try:
                ckpt = torch.load(checkpoint_path, map_location=map_location, weights_only=False)
            except TypeError:
                ckpt = torch.load(checkpoint_path, map_location=map_location)
            if not isinstance(ckpt, dict) or "state_dict" not in ckpt:
                raise ValueError("%s is not a Lightning checkpoint (no 'state_dict')" % (checkpoint_path,))
            hp = dict(ckpt.get("hyper_parameters") or {})
            hp.update(kwargs)
            accepted = set()
            takes_kw = False
            for klass in cls.__mro__:
                init = klass.__dict__.get("__init__")
                if init is None or klass in (nn.Module, object):
                    continue
                sig = inspect.signature(init)
                accepted.update(n for n, prm in sig.parameters.items() if prm.kind in (prm.POSITIONAL_OR_KEYWORD, prm.KEYWORD_ONLY))
                takes_kw = any(prm.kind == prm.VAR_KEYWORD for prm in sig.parameters.values())
                if not takes_kw:
                    break
            model = cls(**{k: v for k, v in hp.items() if k in accepted and k != "self"})
            return _load_reference_state(model, ckpt["state_dict"], strict)


def _load_reference_state(model, state_dict, strict=True):
    """load_state_dict for a checkpoint written by the reference's module tree.  The image encoder lives under ``net_rgb.`` there; a
    model built without one (``net_rgb`` is injected by the caller here) takes everything else and says what it left out; with an
    encoder in place those keys load like any others."""
    sd = dict(state_dict)
    has_encoder = any(True for _ in model.net_rgb.parameters()) or any(True for _ in model.net_rgb.buffers())
    skipped = []
    if not has_encoder:
        skipped = [k for k in sd if k.startswith("net_rgb.")]
        for k in skipped:
            del sd[k]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    if strict and (missing or unexpected):
        raise RuntimeError("load_from_checkpoint: missing keys %s, unexpected keys %s" % (list(missing)[:8], list(unexpected)[:8]))
    if skipped:
        import warnings
        warnings.warn("load_from_checkpoint: %d encoder tensors under 'net_rgb.' were not loaded (no encoder was injected: pass "
                      "net_rgb=<module> to load them)" % len(skipped))
    object.__setattr__(model, "_skipped_checkpoint_keys", skipped)
    return model


class ResnetBlockFC(nn.Module):
    """Parameter container for one block (resnetfc.py:20-52): fc_0, fc_1 with the reference initialisation."""

    def __init__(self, size: int):
        super().__init__()
        self.fc_0 = nn.Linear(size, size)
        self.fc_1 = nn.Linear(size, size)
        nn.init.constant_(self.fc_0.bias, 0.0)
        nn.init.kaiming_normal_(self.fc_0.weight, a=0, mode="fan_in")
        nn.init.constant_(self.fc_1.bias, 0.0)
        nn.init.zeros_(self.fc_1.weight)


class ResnetFC(nn.Module):
    """Parameter container with the reference's names/shapes/init (resnetfc.py:67-131).

    It has no ``forward``: evaluation happens inside the HIP MLP pass.  The shape SceneRF instantiates (``n_blocks=3, d_hidden=512``,
    scenerf.py:100-114) runs on the fused kernels, forward and backward; every other block count / hidden width (the reference class is
    generic: BASELINE.json configs[0] names 1 block x 128) runs in fp32 with one MFMA GEMM per ``nn.Linear`` -- and, since round 6, one per
    gradient: ``scenerf_hip_resnetfc_forward`` under ``no_grad``, ``_forward_train`` / ``_backward`` otherwise.  Assign such a net to
    ``model.mlp`` / ``model.mlp_gaussian`` of a ``precision="fp32"`` model.
    """

    def __init__(self, d_in: int = 42, d_out: int = 4, n_blocks: int = 3, d_latent: int = 2480, d_hidden: int = 512):
        super().__init__()
        if (d_in, d_latent) != (42, 2480) or d_out not in (2, 4):
            raise ValueError("the HIP ray pipeline feeds a ResnetFC(d_in=42, d_latent=2480, d_out in {2,4}) (positional encoding + view direction, "
                             "the five feature maps' 2480 channels)")
        if not (1 <= n_blocks <= 8) or d_hidden < 16 or d_hidden % 16:
            raise ValueError("ResnetFC: n_blocks in 1..8 and d_hidden a multiple of 16 (the MFMA GEMM's K granularity)")
        self.d_in, self.d_out, self.n_blocks, self.d_latent, self.d_hidden = d_in, d_out, n_blocks, d_latent, d_hidden
        self.lin_in = nn.Linear(d_in, d_hidden)
        nn.init.constant_(self.lin_in.bias, 0.0)
        nn.init.kaiming_normal_(self.lin_in.weight, a=0, mode="fan_in")
        self.lin_out = nn.Linear(d_hidden, d_out)
        nn.init.constant_(self.lin_out.bias, 0.0)
        nn.init.kaiming_normal_(self.lin_out.weight, a=0, mode="fan_in")
        self.blocks = nn.ModuleList([ResnetBlockFC(d_hidden) for _ in range(n_blocks)])
        self.lin_z = nn.ModuleList([nn.Linear(d_latent, d_hidden) for _ in range(n_blocks)])
        for i in range(n_blocks):
            nn.init.constant_(self.lin_z[i].bias, 0.0)
            nn.init.kaiming_normal_(self.lin_z[i].weight, a=0, mode="fan_in")

    @property
    def is_standard(self) -> bool:
        """The trunk SceneRF instantiates: the shape the fused kernels (forward + backward) are built for."""
        return self.n_blocks == 3 and self.d_hidden == 512

    def ordered_params(self):
        """Parameters in the order the renderer's pack expects: ``MLP_PARAM_NAMES`` for the standard shape; the same pattern over
        ``n_blocks`` blocks otherwise (lin_in, lin_out, then per block fc_0, fc_1, lin_z: weight, bias)."""
        p = dict(self.named_parameters())
        if self.is_standard:
            return [p[n] for n in MLP_PARAM_NAMES]
        names = ["lin_in.weight", "lin_in.bias", "lin_out.weight", "lin_out.bias"]
        for b in range(self.n_blocks):
            names += ["blocks.%d.fc_0.weight" % b, "blocks.%d.fc_0.bias" % b, "blocks.%d.fc_1.weight" % b, "blocks.%d.fc_1.bias" % b,
                      "lin_z.%d.weight" % b, "lin_z.%d.bias" % b]
        return [p[n] for n in names]

    def forward(self, *a, **k):
        raise RuntimeError("ResnetFC is evaluated by the HIP MLP pass; call SceneRF.render_rays_batch")


class PositionalEncoding(nn.Module):
    """Buffers of the reference encoding (pe.py:13-30); the encoding itself is fused into encode_points."""

    def __init__(self, num_freqs: int = 6, d_in: int = 3, freq_factor: float = np.pi, include_input: bool = True):
        super().__init__()
        if num_freqs != 6 or d_in != 3 or not include_input:
            raise ValueError("the HIP encoder is built for num_freqs=6, d_in=3, include_input=True")
        self.num_freqs, self.d_in, self.include_input = num_freqs, d_in, include_input
        self.freqs = freq_factor * 2.0 ** torch.arange(0, num_freqs)
        self.d_out = num_freqs * 2 * d_in + d_in
        self.register_buffer("_freqs", torch.repeat_interleave(self.freqs, 2).view(1, -1, 1))
        ph = torch.zeros(2 * num_freqs)
        ph[1::2] = np.pi * 0.5
        self.register_buffer("_phases", ph.view(1, -1, 1))


class RaySOM(nn.Module):
    def __init__(self, som_sigma: float):
        super().__init__()
        self.som_sigma = som_sigma


class SphericalMapping(nn.Module):
    """Constants of the sphere projection + ``from_pixels`` for the encoder input (spherical_mapping.py:47-115).

    ``from_pixels`` here serves the *encoder* (one call per image, off the hot path); the per-sample
    spherical indices of the hot path are computed in the encode_points kernel.
    """

    def __init__(self, img_W, img_H, out_img_W, out_img_H, v_angle_max, v_angle_min, h_angle_max, h_angle_min):
        super().__init__()
        self.img_W, self.img_H, self.out_img_W, self.out_img_H = img_W, img_H, out_img_W, out_img_H
        self.v_angle_max, self.v_angle_min = v_angle_max, v_angle_min
        self.h_angle_max, self.h_angle_min = h_angle_max, h_angle_min
        self.h_fov = abs(h_angle_max - h_angle_min)
        self.v_fov = abs(v_angle_max - v_angle_min)

    def _full_grid(self, like: torch.Tensor) -> torch.Tensor:
        """All (u, v) pixel coordinates of the image, built once per (device, dtype): the reference rebuilds the 451k-point meshgrid on
        the CPU and uploads it on every step (spherical_mapping.py:84-90)."""
        key = (like.device, like.dtype)
        cache = self.__dict__.setdefault("_grid_cache", {})
        if key not in cache:
            ys, xs = torch.meshgrid(torch.arange(self.img_H, device=like.device), torch.arange(self.img_W, device=like.device),
                                    indexing="ij")
            cache[key] = torch.stack([xs.reshape(-1), ys.reshape(-1)], 1).to(like.dtype)
        return cache[key]

    def from_pixels(self, inv_K, pix_coords=None):
        """spherical_mapping.py:80-115.  On the GPU: one launch of the library's rule (``scenerf_hip_pixels_to_sphere``, the same
        operation sequence -- torch-CPU's, with SLEEF's 1.0-ULP acos / atan2 -- the renderer uses for its per-sample index, so the map
        the encoder fills and the texels the renderer reads agree).  CPU tensors take the reference's torch ops unchanged."""
        if pix_coords is None:
            pix_coords = self._full_grid(inv_K)
        if inv_K.is_cuda:
            from . import _capi
            lib = _capi.load()
            dev = inv_K.device
            # the pixel list follows the intrinsics' device (a CPU list with CUDA intrinsics would hand the kernel a host pointer, a list
            # on another GPU a foreign one); the launch is issued under that device's guard, on its current stream
            pix = pix_coords.to(device=dev, dtype=torch.float32).contiguous()
            ik = inv_K.to(torch.float32).contiguous()
            M = pix.shape[0]
            idx = torch.empty((M, 2), dtype=torch.int64, device=dev)
            dist = torch.empty((M,), dtype=torch.float32, device=dev)
            if M == 0:              # (the reference returns empty tensors here)
                return pix_coords, idx, dist.type_as(inv_K)
            with torch.cuda.device(dev):
                _capi.check(lib.scenerf_hip_pixels_to_sphere(pix.data_ptr(), ik.data_ptr(), self.v_angle_min, self.v_fov, self.h_angle_min,
                                                             self.h_fov, self.out_img_W, self.out_img_H, M, idx.data_ptr(), dist.data_ptr(),
                                                             torch.cuda.current_stream(dev).cuda_stream), "pixels_to_sphere")
            return pix_coords, idx, dist.type_as(inv_K)
        homo = torch.cat([pix_coords, torch.ones_like(pix_coords[:, :1])], dim=1)
        cam = (inv_K @ homo.T).T
        dist = torch.linalg.norm(cam, ord=2, dim=1)
        v_angle = torch.acos(-cam[:, 1] / dist) / math.pi * 180
        h_angle = 180 - torch.atan2(cam[:, 2], cam[:, 0]) / math.pi * 180
        out = torch.zeros((cam.shape[0], 2)).type_as(cam)
        out[:, 0] = (h_angle - self.h_angle_min) / self.h_fov * (self.out_img_W - 1)
        out[:, 1] = (v_angle - self.v_angle_min) / self.v_fov * (self.out_img_H - 1)
        return pix_coords, torch.round(out).long(), dist


class SceneRF(TrainingMixin, _Base):
    _VARIANT = "kitti"

    def __init__(self, som_sigma, lr=1e-5, weight_decay=0, img_size=(1220, 370), n_rays=1200, max_infer_depth=120,
                 max_sample_depth=100, eval_depth=80, std=2.5, n_gaussians=4, n_pts_uni=32, n_pts_per_gaussian=8,
                 sampling_method="uniform", batch_size=1, add_fov_hor=0, add_fov_ver=0, sphere_H=452, sphere_W=1500,
                 use_color=True, use_reprojection=True, net_rgb: Optional[nn.Module] = None, precision: str = "fp32",
                 device_rng: bool = False):
        super().__init__()
        if sampling_method != "uniform":
            raise ValueError("only sampling_method='uniform' is reachable in the reference (scenerf.py:612)")
        self.use_color, self.use_reprojection = use_color, use_reprojection
        self.lr, self.weight_decay = lr, weight_decay
        self.img_size = img_size
        self.sampling_method = sampling_method
        self.n_rays = n_rays
        self.n_pts_uni, self.n_gaussians, self.n_pts_per_gaussian = n_pts_uni, n_gaussians, n_pts_per_gaussian
        self.std = std
        self.batch_size = batch_size
        self.max_infer_depth, self.max_sample_depth, self.eval_depth = max_infer_depth, max_sample_depth, eval_depth
        self.out_img_W, self.out_img_H = sphere_W, sphere_H
        fov = KITTI_FOV if self._VARIANT == "kitti" else BF_FOV
        self.spherical_mapping = SphericalMapping(
            v_angle_max=fov["v_angle_max"] + add_fov_ver, v_angle_min=fov["v_angle_min"] - add_fov_ver,
            h_angle_max=fov["h_angle_max"] + add_fov_hor, h_angle_min=fov["h_angle_min"] - add_fov_hor,
            img_W=img_size[0], img_H=img_size[1], out_img_W=sphere_W, out_img_H=sphere_H)
        # the encoder stays stock PyTorch (EfficientNet-B7 U-Net in the reference, needs torch.hub); inject it
        self.net_rgb = net_rgb if net_rgb is not None else nn.Module()
        try:
            self.save_hyperparameters(ignore=["net_rgb"])
        except TypeError:
            self.save_hyperparameters()
        self.pe = PositionalEncoding(num_freqs=6, include_input=True)
        self.mlp = ResnetFC(d_in=39 + 3, d_out=4, n_blocks=3, d_hidden=512, d_latent=2480)
        self.mlp_gaussian = ResnetFC(d_in=39 + 3, d_out=2, n_blocks=3, d_hidden=512, d_latent=2480)
        self.ray_som = RaySOM(som_sigma=som_sigma)
        self.render_cfg = RenderConfig(
            img_size=tuple(img_size), sphere_W=sphere_W, sphere_H=sphere_H, add_fov_hor=add_fov_hor,
            add_fov_ver=add_fov_ver, n_pts_uni=n_pts_uni, n_gaussians=n_gaussians,
            n_pts_per_gaussian=n_pts_per_gaussian, max_sample_depth=float(max_sample_depth), std=float(std),
            som_sigma=float(som_sigma), gauss_floor=1.5 if self._VARIANT == "kitti" else 0.5,
            uni_fallback=0 if self._VARIANT == "kitti" else 2,
            precision=precision, device_rng=device_rng, **fov)
        self.render_cfg.validate()
        # optional data-parallel hook (scenerf_amd.dist.allreduce_mean_): called on each MLP's packed gradient buffer, once per
        # render_rays_batch session -- every rank must then open the same sessions per step (dist.verify_step_collectives checks it)
        # and the parameters must not also carry DDP gradient hooks; the once-per-step alternative is dist.GradBucket after backward
        self.grad_sync = None
        # optional early form of the same hook (scenerf_amd.dist.allreduce_mean_async): the radiance MLP's collective then starts
        # before the feature-gradient scatter of a single-chunk (training) step instead of after it
        self.grad_sync_async = None
        # debugging / parity tests: keep the stage intermediates of the last rendered chunk in ``self.last_aux`` (off by default: it
        # pins a chunk's logits / encodings / indices for as long as the model lives)
        self.debug_aux = False
        self.last_aux = None

    # ---- the hot path ---------------------------------------------------------------------------------------
    def _inv_K(self, cam_K: torch.Tensor) -> torch.Tensor:
        """torch.inverse(cam_K) (scenerf.py:400), cached per intrinsics tensor.  Computed by the HOST's LAPACK and uploaded: the GPU's LU
        (rocSOLVER: ~8 tiny launches, not stream-capturable) returns an inverse that differs from the CPU's in the last bit of single
        entries (BundleFusion's -cy/fy: seen as 6 of 108,000 sphere rows off the oracle's), and everything downstream of the sphere index
        is held bit-exact to torch-CPU's operation sequence (csrc/sphere_exact.h).  One small D2H + H2D per NEW intrinsics tensor; the
        cache key is the tensor OBJECT and its version counter (never the address: a new tensor may reuse it), so a loop -- or a
        captured graph -- that passes the same K tensor pays once."""
        # (a few entries: a batch of several images brings one intrinsics tensor per image -- scenerf.py:141-156 -- and alternates them)
        multi = self.__dict__.setdefault("_inv_K_more", {})
        hit = getattr(self, "_inv_K_cache", None)
        if (hit is None or hit[0] is not cam_K) and id(cam_K) in multi and multi[id(cam_K)][0] is cam_K:
            hit = multi[id(cam_K)]
            object.__setattr__(self, "_inv_K_cache", hit)
        if hit is None or hit[0] is not cam_K or hit[1] != cam_K._version:
            if torch.cuda.is_available() and cam_K.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("render_rays_batch: new intrinsics inside a hipGraph capture (their inverse is made on the host: a "
                                   "device-to-host copy cannot be captured) -- render once with this cam_K tensor before capturing")
            inv = torch.inverse(cam_K.detach().to("cpu", torch.float32)).contiguous()   # (torch.inverse hands back a column-major view)
            if hit is not None and hit[0] is cam_K and hit[2].device == cam_K.device:
                # same tensor object, new values (cam_K.copy_(...)): refresh the cached inverse IN PLACE -- a captured step holds its
                # address (GraphedStep.__call__ comes through here when the version counter of its static cam_K has moved)
                hit[2].copy_(inv)
                hit = (cam_K, cam_K._version, hit[2])
            else:
                hit = (cam_K, cam_K._version, inv.to(cam_K.device))
            object.__setattr__(self, "_inv_K_cache", hit)
            if len(multi) >= 8 and id(cam_K) not in multi:
                multi.pop(next(iter(multi)))
            multi[id(cam_K)] = hit
        return hit[2]

    def _device_rng_state(self, dev) -> torch.Tensor:
        """{seed, calls so far, scratch} of the in-kernel sampler noise (``device_rng=True``): one per model and device, kept across calls
        -- a captured hipGraph holds its address and every replay advances the call counter on the device.  The seed comes from torch's
        CPU generator when the state is first needed (``torch.manual_seed`` makes a run repeatable); ``reseed_device_rng`` resets it."""
        st = self.__dict__.setdefault("_rng_states", {})
        # (lane 1: the trainer's metric-only renders when they run on a stream of their own beside the trained renders -- two renders in
        #  flight must not advance one call counter: training.TrainingMixin.overlap_metric_renders)
        lane = int(self.__dict__.get("_rng_lane", 0))
        key = torch.device(dev).index if lane == 0 else (torch.device(dev).index, lane)
        if key not in st:
            st[key] = torch.tensor([int(torch.randint(0, 2 ** 62, (1,)).item()), 0, 0], dtype=torch.int64, device=dev)
        return st[key]

    def reseed_device_rng(self, seed: int) -> None:
        for key, t in self.__dict__.get("_rng_states", {}).items():
            lane = key[1] if isinstance(key, tuple) else 0
            t.copy_(torch.tensor([(int(seed) + 0x9E3779B97F4A7C15 * lane) % (2 ** 62), 0, 0], dtype=torch.int64))

    def render_rays_batch(self, cam_K, T_source2infer, x_rgb: Dict[str, torch.Tensor], depth_window=100,
                          T_cam2velo=None, sampled_pixels=None, ray_batch_size=128, noise=None):
        """scenerf.py:392-471.  ``depth_window`` / ``T_cam2velo`` are accepted and unused, as in the reference.

        ``noise=(noise_u (n,U,1), noise_g (n,G*P))`` optionally injects the sampling noise (tests / replay);
        by default it is drawn like the reference does (rand on device, normal on CPU).
        """
        if sampled_pixels is None:
            raise ValueError("sampled_pixels is required")
        if sampled_pixels.shape[0] == 0:
            raise ValueError("sampled_pixels is empty (the reference fails in torch.cat over zero chunks here, scenerf.py:459-470)")
        self.ray_som  # noqa: B018  (attribute kept for parity with the reference module tree)
        if (self.static_inference and not torch.is_grad_enabled() and sampled_pixels.shape[0] > ray_batch_size and not self.debug_aux
                and self.mlp.is_standard and self.mlp_gaussian.is_standard):
            # the evaluation / reconstruction callers (render_colors.py:114-119, generate_novel_depths.py:116-122: 50k-450k rays in
            # chunks of 4000-8000 under no_grad): static chunk shape, one captured hipGraph per input frame (scenerf_amd/inference.py)
            return self.render_image(cam_K, T_source2infer, x_rgb, sampled_pixels=sampled_pixels, ray_batch_size=ray_batch_size, noise=noise)
        cfg = self.render_cfg
        cfg.som_sigma = float(self.ray_som.som_sigma)
        inv_K = self._inv_K(cam_K)
        rng = self._device_rng_state(sampled_pixels.device) if (cfg.device_rng and noise is None) else None
        # Inside the trainer's forward (training.TrainingMixin._params_fixed: no optimizer step in there, ONE backward over the returned
        # total) the S trained renders of an image share one session, and so do its S metric-only renders: maps converted once, operands
        # packed once, and -- the part the caches above cannot give -- ONE set of gradient accumulators: every source frame's backward adds
        # into the same map-gradient and parameter-gradient sinks, handed to autograd once (no per-source zero fill / transpose back /
        # unpack, no AccumulateGrad additions of 420 MB maps and 40 parameter tensors; N > 1: one gradient all-reduce per image instead
        # of one per source frame).  A session renders chunks of any pose (RenderSession.render_chunk), which is all this needs
        shared, skey, sess = self.__dict__.get("_image_sessions"), None, None
        if shared is not None and not self.debug_aux and self.share_image_sessions:
            skey = self._session_key(cfg, x_rgb, rng, sampled_pixels.device)
            sess = shared.get(skey)
        reused = sess is not None
        if sess is None:
            sess = RenderSession(cfg, x_rgb, self.mlp.ordered_params(), self.mlp_gaussian.ordered_params(),
                                 grad_sync=self.grad_sync, grad_sync_async=self.grad_sync_async, debug_aux=self.debug_aux, rng=rng,
                                 events=self.__dict__.setdefault("_step_events", {}),
                                 convert_cache=self.__dict__.setdefault("_convert_cache", {}) if self.cache_converted_maps else None,
                                 pack_cache=self.__dict__.get("_pack_cache"))
            if skey is not None:
                shared[skey] = sess
        outs, auxs = [], []
        n = sampled_pixels.shape[0]
        # training (scenerf.py:262-275), one chunk in the whole session: lets the head's gradient all-reduce start early
        sess.mlpg.single_chunk = (n <= ray_batch_size) and not reused
        for s in range(0, n, ray_batch_size):
            e = min(s + ray_batch_size, n)
            nu = noise[0][s:e] if noise is not None else None
            ng = noise[1][s:e] if noise is not None else None
            outs.append(sess.render_chunk(sampled_pixels[s:e], cam_K, inv_K, T_source2infer, nu, ng))
            if self.debug_aux:
                auxs.append(dict(sess.last_aux))
        if self.debug_aux:   # stage intermediates of all chunks, concatenated ray-major like the outputs
            object.__setattr__(self, "last_aux", {k: (torch.cat([a[k] for a in auxs], dim=0) if auxs[0][k] is not None else None)
                                                  for k in auxs[0]})
        if len(outs) == 1:
            ret = outs[0]
        else:
            ret = {k: torch.cat([o[k] for o in outs], dim=0) for k in outs[0]}
        return {k: ret[k] for k in OUTPUT_KEYS}

    share_image_sessions = True     # see render_rays_batch: only ever in effect inside training.TrainingMixin.forward

    def _session_key(self, cfg, x_rgb, rng, dev):
        from . import _capi
        import ctypes
        cid = ctypes.c_ulonglong(0)
        _capi.check(_capi.load().scenerf_hip_stream_capture_id(torch.cuda.current_stream(dev).cuda_stream, ctypes.byref(cid)), "stream_capture_id")
        from .renderer import HWC
        ts = [(k, v.t if isinstance(v, HWC) else v) for k, v in sorted(x_rgb.items())]
        maps = tuple((k, id(v), v._version, v.data_ptr()) for k, v in ts)
        return (torch.is_grad_enabled(), float(cfg.som_sigma), cid.value, None if rng is None else rng.data_ptr(), maps)

    # ---- full-frame inference (BASELINE.json configs[4]) -------------------------------------------------------------------------
    # keep the (H,W,C) copies of the last image's contiguous (C,H,W) maps between calls (keyed by tensor identity + version counter): the S
    # source frames of one image convert once.  Costs the copies' memory (210 MB in bf16 at KITTI) until the next image replaces them
    cache_converted_maps = True
    static_inference = True     # no_grad multi-chunk render_rays_batch calls go through render_image (padded static chunks + hipGraph)
    inference_graph = True      # replay a captured hipGraph per chunk (False: the same static chunks, launched eagerly)

    def render_image(self, cam_K, T_source2infer, x_rgb: Dict[str, torch.Tensor], sampled_pixels=None, stride: int = 1,
                     ray_batch_size: int = 4096, keys=None, use_graph: Optional[bool] = None, noise=None):
        """Render a whole pixel set of one pose under ``no_grad``: every ``stride``-th pixel of the image in the reference scripts'
        order when ``sampled_pixels`` is None (render_colors.py:102-111).  The tail chunk is padded to ``ray_batch_size`` rays so that
        every chunk has one static shape; one chunk is captured into a hipGraph per input frame and replayed for all chunks and all
        poses rendered from that frame.  Sampling noise follows ``device_rng`` as in the chunk loop.  Returns the dict of ``render_rays_batch``
        (``keys`` selects a subset: the (n, N) outputs of a 451,400-ray frame at N = 512 are 0.9 GB each)."""
        from .inference import ImageRenderer, pixel_grid
        if not (self.mlp.is_standard and self.mlp_gaussian.is_standard):
            raise NotImplementedError("render_image needs the 3 x 512 ResnetFC trunks (its static-chunk engine re-packs the fused kernels' operand "
                                      "layouts in place); a model with another ResnetFC shape renders through render_rays_batch under no_grad")
        if sampled_pixels is None:
            sampled_pixels = pixel_grid(tuple(self.img_size), stride, x_rgb["1_1"].device)
        if sampled_pixels.shape[0] == 0:
            raise ValueError("sampled_pixels is empty")
        # default: a graph once a call is long enough to pay for the capture ("auto"); use_graph=True/False forces either way
        graph = ("auto" if self.inference_graph else False) if use_graph is None else bool(use_graph)
        key = (int(ray_batch_size), graph, tuple(keys) if keys is not None else None, repr(self.render_cfg), float(self.ray_som.som_sigma))
        eng = getattr(self, "_image_renderer", None)
        # an engine is reused only for the SAME map and parameter objects at the same addresses (it holds strong references, so neither
        # ids nor addresses can be recycled under it); their VALUES are re-read on every call (ImageRenderer.refresh), so writes that
        # no version counter sees (p.data.copy_(ema), load_state_dict) cannot be served stale
        if eng is None or eng[0] != key or not eng[1].matches(self, x_rgb):
            eng = (key, ImageRenderer(self, x_rgb, chunk=int(ray_batch_size), use_graph=graph, keys=keys))
            object.__setattr__(self, "_image_renderer", eng)     # one frame's engine at a time (its graph pins a chunk's buffers)
        return eng[1].render(cam_K, T_source2infer, sampled_pixels, noise=noise)

    def release_inference_engine(self) -> None:
        """Drop the cached full-frame engine (its hipGraph pins one chunk's buffers and the converted maps of the last frame)."""
        object.__setattr__(self, "_image_renderer", None)

    def configure_optimizers(self):
        optimizer = torch.optim.AdamW(self.parameters(), lr=self.lr, weight_decay=self.weight_decay)
        scheduler = torch.optim.lr_scheduler.ExponentialLR(optimizer, gamma=0.95)
        return [optimizer], [scheduler]


class SceneRFBundleFusion(BundleFusionTrainingMixin, SceneRF):
    """scenerf/models/scenerf_bf.py: same kernels, indoor constants (FOV, +0.5 floors, loss weights x5 / x0.1) and the
    BundleFusion batch layout in ``forward`` (``cam_K_depth``, ``source_depths``, ``n_rays // sample_grid_size**2`` rays per source:
    scenerf_amd.training.BundleFusionTrainingMixin)."""
    _VARIANT = "bf"

    def __init__(self, som_sigma, lr=1e-4, weight_decay=0, img_size=(640, 480), sample_grid_size=2, n_rays=1000,
                 max_sample_depth=12, eval_depth=10, std=0.2, n_gaussians=4, n_pts_uni=32, n_pts_per_gaussian=8,
                 smooth_loss_weight=0, sampling_method="uniform", batch_size=1, net_2d="b7", add_fov_hor=0, add_fov_ver=0,
                 sphere_H=480, sphere_W=640, use_color=True, use_reprojection=True, **kw):
        if net_2d != "b7":
            raise ValueError("net_2d not found")
        super().__init__(som_sigma, lr=lr, weight_decay=weight_decay, img_size=img_size, n_rays=n_rays,
                         max_sample_depth=max_sample_depth, eval_depth=eval_depth, std=std, n_gaussians=n_gaussians,
                         n_pts_uni=n_pts_uni, n_pts_per_gaussian=n_pts_per_gaussian, sampling_method=sampling_method,
                         batch_size=batch_size, add_fov_hor=add_fov_hor, add_fov_ver=add_fov_ver, sphere_H=sphere_H,
                         sphere_W=sphere_W, use_color=use_color, use_reprojection=use_reprojection, **kw)
        self.sample_grid_size = sample_grid_size
        self.smooth_loss_weight = smooth_loss_weight
