"""Deterministic synthetic inputs for the ray-rendering hot path.

Everything here is generated with numpy's PCG64 ``default_rng`` (not torch's RNG) so the
golden-fixture generator (tests/golden/make_golden.py, which imports the real reference),
the CPU oracle, the GPU parity tests and bench.py all see bit-identical inputs on any box.

Shapes follow SURVEY.md §8(d):
  * KITTI intrinsics from reference scenerf/scripts/determine_angles.py:12-14
  * feature-map channel counts 80/160/320/640/1280 and spatial sizes ``round(size/scale)``
    from reference scenerf/models/unet2d_sphere.py:139
  * relative poses ``Ry(angle) @ trans_z(step)`` as reference scenerf/models/utils.py:29-49
  * stride-2 pixel grid + random subset as reference scenerf/models/scenerf.py:253-264
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch

FEAT_SCALES = (1, 2, 4, 8, 16)
FEAT_CHANNELS = (80, 160, 320, 640, 1280)  # sums to d_latent = 2480


def kitti_cam_K() -> torch.Tensor:
    return torch.tensor(
        [[707.0912, 0.0, 601.8873], [0.0, 707.0912, 183.1104], [0.0, 0.0, 1.0]], dtype=torch.float32
    )


def bundlefusion_cam_K() -> torch.Tensor:
    # not in the reference tree (read from info.txt); value derived from the FOV constants, SURVEY §8(d)
    return torch.tensor([[583.0, 0.0, 320.0], [0.0, 583.0, 240.0], [0.0, 0.0, 1.0]], dtype=torch.float32)


def rel_pose(step_m: float, angle_deg: float = 0.0) -> torch.Tensor:
    """T_source2infer = Ry(angle) @ translate_z(step) (float32 4x4)."""
    rad = angle_deg / 180.0 * math.pi
    t = torch.eye(4)
    t[2, 3] += step_m
    r = torch.eye(4)
    r[:3, :3] = torch.tensor(
        [[math.cos(rad), 0.0, math.sin(rad)], [0.0, 1.0, 0.0], [-math.sin(rad), 0.0, math.cos(rad)]]
    )
    return r @ t


def feature_map_shapes(sphere_W: int, sphere_H: int) -> Dict[str, Tuple[int, int, int]]:
    """(C, H, W) per key "1_s"; spatial sizes use Python round() like the reference decoder."""
    out = {}
    for s, c in zip(FEAT_SCALES, FEAT_CHANNELS):
        out["1_%d" % s] = (c, round(sphere_H / s), round(sphere_W / s))
    return out


def feature_maps(sphere_W: int, sphere_H: int, seed: int, smooth: bool = False, amp: float = 0.5) -> Dict[str, torch.Tensor]:
    """Synthetic encoder output: dict "1_1".."1_16" of float32 (C, H, W) maps.

    ``smooth=True`` low-pass filters each map (3 box passes) so neighbouring texels are
    correlated like a real decoder output; ``False`` gives white noise (worst case for
    index-rounding sensitivity).
    """
    rng = np.random.default_rng(seed)
    maps = {}
    for key, shp in feature_map_shapes(sphere_W, sphere_H).items():
        a = rng.standard_normal(shp, dtype=np.float32) * np.float32(amp)
        t = torch.from_numpy(a)
        if smooth:
            k = torch.ones(1, 1, 5, 5) / 25.0
            t4 = t.unsqueeze(1)
            for _ in range(3):
                t4 = torch.nn.functional.conv2d(t4, k, padding=2)
            t = (t4.squeeze(1) * 6.0).contiguous()
        maps[key] = t
    return maps


# names and shapes of reference ResnetFC(d_in=42, d_out, n_blocks=3, d_hidden=512, d_latent=2480)
def mlp_param_shapes(d_out: int, d_in: int = 42, d_hidden: int = 512, d_latent: int = 2480, n_blocks: int = 3):
    shapes = {"lin_in.weight": (d_hidden, d_in), "lin_in.bias": (d_hidden,),
              "lin_out.weight": (d_out, d_hidden), "lin_out.bias": (d_out,)}
    for b in range(n_blocks):
        shapes["blocks.%d.fc_0.weight" % b] = (d_hidden, d_hidden)
        shapes["blocks.%d.fc_0.bias" % b] = (d_hidden,)
        shapes["blocks.%d.fc_1.weight" % b] = (d_hidden, d_hidden)
        shapes["blocks.%d.fc_1.bias" % b] = (d_hidden,)
        shapes["lin_z.%d.weight" % b] = (d_hidden, d_latent)
        shapes["lin_z.%d.bias" % b] = (d_hidden,)
    return shapes


def mlp_state(seed: int, d_out: int, d_in: int = 42, d_hidden: int = 512, d_latent: int = 2480,
              n_blocks: int = 3, out_scale: float = 1.0) -> Dict[str, torch.Tensor]:
    """Random ResnetFC state_dict (reference key names).

    Kaiming-like weights (std sqrt(2/fan_in)) everywhere except ``fc_1`` which gets
    std 0.02 (the reference zero-initialises it, which would make the blocks identities
    and hide errors); small non-zero biases so the bias paths are exercised.
    """
    rng = np.random.default_rng(seed)
    state = {}
    for name, shp in mlp_param_shapes(d_out, d_in, d_hidden, d_latent, n_blocks).items():
        if name.endswith("bias"):
            a = rng.standard_normal(shp, dtype=np.float32) * np.float32(0.05)
        else:
            fan_in = shp[1]
            std = 0.02 if ".fc_1." in name else math.sqrt(2.0 / fan_in)
            if name.startswith("lin_in"):
                # raw xyz up to ~100 m enters lin_in; keep pre-activations O(1)
                std = std * 0.1
            if name.startswith("lin_out"):
                std = std * out_scale
            a = rng.standard_normal(shp, dtype=np.float32) * np.float32(std)
        state[name] = torch.from_numpy(a)
    if d_out == 4:
        # keep densities moderate (sigma = softplus(out-1) ~ 0.05) so transmittance decays over
        # tens of samples instead of saturating at the first one: exercises the whole scan.
        state["lin_out.weight"] = state["lin_out.weight"] * 0.3
        state["lin_out.bias"][3] -= 2.0
    return state


def stride2_pixels(img_size: Tuple[int, int], n_rays: int, seed: int) -> torch.Tensor:
    """(n_rays, 2) float32 (u, v) pixels: stride-2 grid + random subset without replacement."""
    W, H = img_size
    xs = np.arange(0, W, 2, dtype=np.float32)
    ys = np.arange(0, H, 2, dtype=np.float32)
    gx, gy = np.meshgrid(xs, ys, indexing="ij")
    grid = np.stack([gx, gy], axis=2).reshape(-1, 2)
    rng = np.random.default_rng(seed)
    idx = rng.permutation(grid.shape[0])[:n_rays]
    if n_rays > grid.shape[0]:
        idx = rng.integers(0, grid.shape[0], size=n_rays)
    return torch.from_numpy(grid[idx].copy())


def sampling_noise(n_rays: int, n_pts_uni: int, n_gauss_pts: int, seed: int):
    """(noise_u (R,U,1) in [0,1), noise_g (R,G*P) ~ N(0,1)), float32."""
    rng = np.random.default_rng(seed)
    nu = rng.random((n_rays, n_pts_uni, 1), dtype=np.float32)
    ng = rng.standard_normal((n_rays, n_gauss_pts), dtype=np.float32)
    return torch.from_numpy(nu), torch.from_numpy(ng)
