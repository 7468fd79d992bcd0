"""CPU ORACLE for the SceneRF ray-rendering hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain fp32 PyTorch-on-CPU, the algorithm of the reference hot path
(astra-vision/SceneRF, ``scenerf/models``).  It is the checker for the HIP kernels: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product (``scenerf_amd``) never imports or falls back to it.

Pinning: the reference ships no tests / golden vectors, so the pins are
minted from the reference itself: ``tests/golden/make_golden.py`` imports the unmodified
reference modules from ``/root/reference`` (with a 10-line pytorch_lightning shim), runs
them on seeded synthetic inputs with the sampling noise injected, and stores the outputs
and gradients as ``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` checks this file
against every one of them.  Third-party arithmetic on the path lives in PyTorch ATen
(grid_sampler_2d, addmm, cumprod, sort, sin/acos/atan2, softplus) -- reference pins
torch 1.7.1; we run torch 2.10 CPU, no behavioural difference observed.

Differences from the reference that are *interface only* (never arithmetic):
  * sampling noise is an explicit argument (the reference draws ``torch.rand_like`` on
    device, utils.py:84, and ``torch.normal`` on CPU, utils.py:208-211);
  * pure functions instead of LightningModule methods; per-ray intermediates returned too.

Each function cites the reference file:line it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

FEAT_SCALES = (1, 2, 4, 8, 16)


@dataclass
class OracleConfig:
    """Constants that reach the hot path (reference scenerf.py:23-116 / scenerf_bf.py)."""
    img_size: Tuple[int, int] = (1220, 370)
    sphere_W: int = 1500
    sphere_H: int = 452
    # SphericalMapping FOV constants, scenerf.py:83-88 (+ add_fov from the CLI)
    v_angle_max: float = 104.7294
    v_angle_min: float = 75.4815
    h_angle_max: float = 131.1128
    h_angle_min: float = 49.5950
    add_fov_hor: float = 0.0
    add_fov_ver: float = 0.0
    n_pts_uni: int = 32
    n_gaussians: int = 4
    n_pts_per_gaussian: int = 8
    max_sample_depth: float = 100.0
    std: float = 2.5
    som_sigma: float = 2.0
    gauss_floor: float = 1.5       # scenerf.py:591-594 (+1.5); scenerf_bf.py:606-608 (+0.5)
    kl_std_floor: float = 1.5      # ray_som_kl.py:83 (same in BF)
    uni_fallback: int = 0          # uniform samples drawn when n_pts_uni == 0: scenerf_bf.py:623-626 substitutes 2; scenerf.py has no substitute
    # The arithmetic behind the sphere index (ray direction -> sample point -> projected pixel -> angles -> round()).  "torch": the
    # reference's calls as THIS host executes them.  That is not one function of the inputs: `torch.acos` is MKL VML's vmsAcos(HA), whose last
    # bit depends on the instruction set MKL dispatches to; `A @ x.T` is MKL's sgemm, whose summation order depends on the CPU vendor and the
    # operand layout (on the AMD EPYC hosts of the MI355X boxes `K @ pts.T` differs from the same call on the Intel build container in a
    # third of its elements: profiles/r05_oracle_vs_host_epyc.txt); `torch.atan2` sends the last n mod 32 elements of every OpenMP chunk
    # through the C library instead of SLEEF (oracle/sleef_acos.py, tools/sleef_check/).  "pinned": the rule this project fixes (DESIGN.md
    # section 2) -- 3x3 / 4x4 products as k-ordered fma chains (what MKL does on the Intel container the golden vectors were minted on),
    # acos / atan2 = torch's own SLEEF u10 build; everything else (division, 2-norm, scaling, round) is ATen's and already host-independent.
    # The GPU parity tests select "pinned" and then require EVERY sphere index, sample point and ray direction to be equal, bit for bit.
    index_rule: str = "torch"

    @property
    def fov(self):
        v_max = self.v_angle_max + self.add_fov_ver
        v_min = self.v_angle_min - self.add_fov_ver
        h_max = self.h_angle_max + self.add_fov_hor
        h_min = self.h_angle_min - self.add_fov_hor
        return v_min, abs(v_max - v_min), h_min, abs(h_max - h_min)

    @property
    def n_samples(self):
        if self.n_pts_uni > 0:
            return self.n_pts_uni + self.n_gaussians * self.n_pts_per_gaussian
        if self.n_pts_per_gaussian == 1:       # scenerf.py:647-650: the uniform samples alone
            return self.uni_fallback
        return self.n_gaussians * self.n_pts_per_gaussian

    @staticmethod
    def kitti(**kw):
        d = dict(add_fov_hor=20.0, add_fov_ver=8.0, std=2.0, som_sigma=2.0)  # train_kitti.py CLI defaults
        d.update(kw)
        return OracleConfig(**d)

    @staticmethod
    def bundlefusion(**kw):
        d = dict(img_size=(640, 480), sphere_W=960, sphere_H=720, v_angle_max=112.2911, v_angle_min=67.6248,
                 h_angle_max=118.6861, h_angle_min=61.2383, add_fov_hor=14.0, add_fov_ver=11.0,
                 max_sample_depth=12.0, std=0.1, som_sigma=0.02, gauss_floor=0.5, uni_fallback=2)
        d.update(kw)
        return OracleConfig(**d)


# ----------------------------------------------------------------------------- geometry
def _homog(pix: torch.Tensor) -> torch.Tensor:
    return torch.cat([pix, torch.ones_like(pix[:, :1])], dim=1)


class _PinnedMatvec(torch.autograd.Function):
    """(A @ X.T).T with every output element the k-ordered fma chain fma(a_k, x_k, ... fma(a_1, x_1, a_0 * x_0)) (oracle/sleef_shim.c);
    the gradient w.r.t. X is the plain product (A is a constant of the path)."""

    @staticmethod
    def forward(ctx, A, X):
        import sleef_acos          # (oracle/ is on sys.path: the oracle modules are imported top-level)
        ctx.save_for_backward(A)
        return sleef_acos.matvec_fma(A, X)

    @staticmethod
    def backward(ctx, g):
        (A,) = ctx.saved_tensors
        return None, g @ A


def _matvec(A: torch.Tensor, X: torch.Tensor, rule: str = "torch") -> torch.Tensor:
    """(A @ X.T).T for X (M, k): the reference's call (rule "torch": whatever this host's BLAS makes of it) or the pinned k-ordered fma
    chain (OracleConfig.index_rule)."""
    if rule == "pinned":
        return _PinnedMatvec.apply(A.detach().float().contiguous(), X.float())
    if rule != "torch":
        raise ValueError("index_rule %r" % (rule,))
    return (A @ X.T).T


def ray_directions(pixels: torch.Tensor, inv_K: torch.Tensor, rule: str = "torch"):
    """utils.py:177-182 (unit dirs) and utils.py:131-135,170 (un-normalised dir)."""
    d = _matvec(inv_K[:3, :3], _homog(pixels), rule)
    return d, F.normalize(d, dim=1)


def to_frame(pts: torch.Tensor, T: torch.Tensor, rule: str = "torch") -> torch.Tensor:
    """utils.py:161-166 / 268-279: homogeneous 4x4 transform of (M,3) points."""
    h = torch.cat([pts, torch.ones(pts.shape[0], 1, dtype=pts.dtype, device=pts.device)], dim=1).float()
    return _matvec(T, h, rule)[:, :3]


def uniform_distances(n_rays: int, U: int, D: float, noise_u: torch.Tensor) -> torch.Tensor:
    """utils.py:75-90: linspace(0.2, D, U) + rand * (D-0.2)/U."""
    step = (D - 0.2) / U
    base = torch.linspace(0.2, D, steps=U, device=noise_u.device).reshape(1, U, 1).expand(n_rays, -1, -1)
    return (base + noise_u * step).squeeze(-1)


def gaussian_anchor_distances(cfg: OracleConfig) -> torch.Tensor:
    """scenerf.py:554-560."""
    step = cfg.max_sample_depth * 1.0 / cfg.n_gaussians
    return torch.linspace(step / 2, cfg.max_sample_depth - step / 2, steps=cfg.n_gaussians)


def project_to_pixels(pts: torch.Tensor, K: torch.Tensor, rule: str = "torch") -> torch.Tensor:
    """utils.py:298-315: K @ p, perspective divide where z>0, else (-1,-1)."""
    h = _matvec(K, pts, rule)
    ok = h[:, 2] > 0
    pix = torch.full((pts.shape[0], 2), -1.0, device=pts.device)
    pix = torch.where(ok[:, None], h[:, :2] / h[:, 2:3], pix)
    return pix


def sphere_coords(pix: torch.Tensor, inv_K: torch.Tensor, cfg: OracleConfig, return_float: bool = False):
    """spherical_mapping.py:80-115: pixel -> unit-depth cam point -> (acos, atan2) -> rounded sphere pixel."""
    v_min, v_fov, h_min, h_fov = cfg.fov
    c = _matvec(inv_K, _homog(pix), cfg.index_rule)
    c = torch.ones(pix.shape[0], device=pix.device).view(-1, 1) * c
    x, y, z = c[:, 0], c[:, 1], c[:, 2]
    dist = torch.linalg.norm(c, ord=2, dim=1)
    if cfg.index_rule == "pinned":
        import sleef_acos          # (oracle/ is on sys.path: the oracle modules are imported top-level)
        v_angle = sleef_acos.acos(-y / dist) / math.pi * 180
        # (torch.atan2 is already SLEEF where ATen runs its AVX2 / AVX-512 kernels; calling the routine by name makes the rule
        # independent of ATEN_CPU_CAPABILITY)
        h_angle = 180 - sleef_acos.atan2(z.contiguous(), x.contiguous()) / math.pi * 180
    elif cfg.index_rule == "torch":
        v_angle = torch.acos(-y / dist) / math.pi * 180
        h_angle = 180 - torch.atan2(z, x) / math.pi * 180
    else:
        raise ValueError("index_rule %r" % (cfg.index_rule,))
    out = torch.zeros((pix.shape[0], 2), device=pix.device)
    out[:, 0] = (h_angle - h_min) / h_fov * (cfg.sphere_W - 1)
    out[:, 1] = (v_angle - v_min) / v_fov * (cfg.sphere_H - 1)
    idx = torch.round(out).long()
    if return_float:
        return idx, out
    return idx


def positional_encoding(x: torch.Tensor, num_freqs: int = 6) -> torch.Tensor:
    """pe.py:13-43: [x, sin(f0 x), sin(f0 x + pi/2), sin(f1 x), ...], f_k = pi 2^k, (M,3)->(M,39)."""
    freqs = math.pi * 2.0 ** torch.arange(0, num_freqs, device=x.device)
    f = torch.repeat_interleave(freqs, 2).view(1, -1, 1)
    ph = torch.zeros(2 * num_freqs, device=x.device)
    ph[1::2] = math.pi * 0.5
    e = x.unsqueeze(1).repeat(1, num_freqs * 2, 1)
    e = torch.sin(torch.addcmul(ph.view(1, -1, 1), e, f)).view(x.shape[0], -1)
    return torch.cat((x, e), dim=-1)


def gather_features(fmap: torch.Tensor, idx: torch.Tensor, div_wh: Tuple[int, int]) -> torch.Tensor:
    """utils.py:232-247: bilinear grid_sample (align_corners=False, zeros) at idx/div*2-1. (M, C)."""
    g = (idx / torch.tensor(div_wh, device=idx.device).type_as(idx).reshape(1, 2)) * 2 - 1
    out = F.grid_sample(fmap.unsqueeze(0), g.reshape(1, 1, -1, 2), align_corners=False,
                        mode="bilinear", padding_mode="zeros")
    return out.reshape(out.shape[1], -1).T


def point_inputs(pts: torch.Tensor, viewdir_rows: torch.Tensor, x_rgb: Dict[str, torch.Tensor],
                 K: torch.Tensor, cfg: OracleConfig, idx_use: Optional[torch.Tensor] = None):
    """scenerf.py:505-531: (M,3) infer-frame points -> x_in (M, 2480+39+3) and the sphere indices.
    ``idx_use`` (parity tests only, see render_chunk): gather at these indices instead of the ones computed here (still returned)."""
    pix = project_to_pixels(pts, K, cfg.index_rule)
    idx = sphere_coords(pix, torch.inverse(K), cfg)
    use = idx if idx_use is None else idx_use.to(idx.dtype).reshape(idx.shape)
    pe = positional_encoding(pts)
    feats = [gather_features(x_rgb["1_1"], use, (cfg.sphere_W, cfg.sphere_H))]
    for s in FEAT_SCALES[1:]:
        feats.append(gather_features(x_rgb["1_%d" % s], use, (cfg.sphere_W // s, cfg.sphere_H // s)))
    return torch.cat(feats + [pe, viewdir_rows], dim=-1), idx


# ----------------------------------------------------------------------------- MLP
def resnetfc_forward(p: Dict[str, torch.Tensor], zx: torch.Tensor, d_latent: int = 2480, n_blocks: Optional[int] = None,
                     keep: Optional[dict] = None) -> torch.Tensor:
    """resnetfc.py:133-164 (+ block :54-63): lin_in; per block: += lin_z(z); x + fc_1(relu(fc_0(relu(x)))); lin_out(relu).
    ``n_blocks`` defaults to what the state dict holds (3 for SceneRF; 1 for the 4-layer plumbing config of BASELINE configs[0])."""
    if n_blocks is None:
        n_blocks = sum(1 for k in p if k.startswith("blocks.") and k.endswith(".fc_0.weight"))
    z, x = zx[..., :d_latent], zx[..., d_latent:]
    h = F.linear(x, p["lin_in.weight"], p["lin_in.bias"])
    for b in range(n_blocks):
        h = h + F.linear(z, p["lin_z.%d.weight" % b], p["lin_z.%d.bias" % b])
        if keep is not None:
            keep["h%d" % b] = h
        net = F.linear(F.relu(h), p["blocks.%d.fc_0.weight" % b], p["blocks.%d.fc_0.bias" % b])
        if keep is not None:
            keep["n%d" % b] = net
        h = h + F.linear(F.relu(net), p["blocks.%d.fc_1.weight" % b], p["blocks.%d.fc_1.bias" % b])
    if keep is not None:
        keep["h%d" % n_blocks] = h
    return F.linear(F.relu(h), p["lin_out.weight"], p["lin_out.bias"])


def density_activation(logit: torch.Tensor) -> torch.Tensor:
    """scenerf.py:473-481: softplus(beta=1)(x - 1)."""
    return F.softplus(logit - 1, beta=1)


# ----------------------------------------------------------------------------- compositing
def composite(density, dist, z, colors):
    """scenerf.py:704-748 (render_depth_and_color). dist is clamped at 0 *in place* like the reference."""
    dist[dist < 0] = 0
    deltas = torch.zeros_like(dist)
    deltas[:, 0] = dist[:, 0]
    deltas[:, 1:] = dist[:, 1:] - dist[:, :-1]
    alphas = 1 - torch.exp(-deltas * density)
    shifted = torch.cat([torch.ones_like(alphas[:, :1]), 1 - alphas + 1e-10], -1)
    T = torch.cumprod(shifted, -1)
    weights = alphas * T[:, :-1]
    depth = torch.sum(weights * z, -1)
    color = torch.sum(weights.unsqueeze(-1) * colors, -2)
    closest, idx = torch.min(torch.abs(depth.unsqueeze(-1) - z), dim=1)
    w_at = torch.gather(weights, 1, idx.unsqueeze(-1)).squeeze(-1)
    return dict(alphas=alphas, weights=weights, depth=depth, color=color, closest=closest,
                closest_idx=idx, weights_at_depth=w_at)


# ----------------------------------------------------------------------------- RaySOM
def ray_som_kl(means, stds, dist, alphas, som_sigma: float, kl_std_floor: float = 1.5, bmu_use=None, mask_use=None, info=None):
    """ray_som_kl.py:10-87: SOM update of G gaussians per ray from (sorted dist, alphas) -> KL(pred || SOM).

    ``bmu_use`` (R, N) / ``mask_use`` (R, G), parity tests only: the two DISCRETE choices of this function -- the best-matching unit
    per sample (an argmax over values that tie at the additive floors, :46-52) and the thresholded update mask (:66-70) -- are taken
    from the argument instead of computed here; the ones computed here are still returned / recorded.  ``info`` (dict) receives how
    close each of those choices was: ``bmu_margin`` (R, N) = (best - second best) / best of the argmax, ``mask_margin`` (R, G) =
    distance of the closest thresholded quantity from its threshold, and the own ``mask``.  None / None is the reference's
    arithmetic, unchanged."""
    m = means.detach()
    s = stds.detach()
    d = dist.detach()
    R, G = m.shape
    Np = d.shape[1]
    gap = torch.abs(m.unsqueeze(1) - d.unsqueeze(-1))                       # R, N, G
    nb = torch.zeros(R, G, G).type_as(m)
    for c2 in range(G):
        for c1 in range(G):
            nb[:, c2, c1] = torch.exp(-(m[:, c2] - m[:, c1]) ** 2 / (2 * som_sigma ** 2))
    p12 = nb / nb.sum(dim=2, keepdim=True)
    var = s ** 2
    pz1 = (torch.exp(-gap ** 2 / (2 * var.unsqueeze(1))) / (math.sqrt(2 * math.pi) * s.detach().unsqueeze(1))) + 1e-5
    dens = alphas + 1e-8
    pz1 = pz1 * dens.unsqueeze(-1) + 1e-8
    tmp = pz1.reshape(R, Np, 1, G) * p12.unsqueeze(1) + 1e-8
    pz2 = tmp.sum(-1)
    pbest, bmu = pz2.max(dim=2)
    bmu_own = bmu
    if info is not None:
        top2 = torch.topk(pz2, 2, dim=2).values
        info["bmu_margin"] = (top2[:, :, 0] - top2[:, :, 1]) / top2[:, :, 0]
    if bmu_use is not None:
        bmu = bmu_use.to(bmu.dtype).reshape(bmu.shape)
        pbest = torch.gather(pz2, 2, bmu.unsqueeze(-1)).squeeze(-1)
    new_m = torch.zeros_like(m)
    new_v = torch.zeros_like(s)
    for r in range(G):
        relw = torch.gather(nb[:, r, :], 1, bmu)
        w = relw * pz1[:, :, r] / pbest + 1e-5
        new_m[:, r] = (w * d).sum(dim=1) / (w.sum(dim=1))
        new_v[:, r] = (w * (d - new_m[:, r].unsqueeze(-1)) ** 2).sum(dim=1) / w.sum(dim=1)
    mean_diff = torch.abs(m - new_m)
    std_diff = torch.abs(torch.sqrt(var) - torch.sqrt(new_v))
    mask = ((mean_diff > 0.1) & (new_v > 0)) * ((std_diff > 0.1) & (new_v > 0))
    if info is not None:
        info["mask"] = mask.clone()
        info["mask_margin"] = torch.minimum(torch.minimum((mean_diff - 0.1).abs(), (std_diff - 0.1).abs()), new_v.abs())
    if mask_use is not None:
        mask = mask_use.to(mask.dtype).reshape(mask.shape)
    s2 = torch.sqrt(new_v).detach()
    m2 = new_m.detach()
    s2[s2 < kl_std_floor] = kl_std_floor
    kl = torch.log(s2 / stds + 1e-8) + (stds ** 2 + (means - m2) ** 2) / (2 * s2 ** 2) - 0.5
    loss = (kl * mask).mean(1)
    return loss, new_m, new_v, bmu_own


# ----------------------------------------------------------------------------- the chunk
def render_chunk(cfg: OracleConfig, mlp: Dict[str, torch.Tensor], mlp_gaussian: Dict[str, torch.Tensor],
                 cam_K: torch.Tensor, T_source2infer: torch.Tensor, x_rgb: Dict[str, torch.Tensor],
                 pixels: torch.Tensor, noise_u: torch.Tensor, noise_g: torch.Tensor,
                 keep_intermediates: bool = False, head_offsets: Optional[torch.Tensor] = None,
                 sphere_idx: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                 som_choices: Optional[Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]] = None) -> Dict[str, torch.Tensor]:
    """scenerf.py:598-700 (batchify_depth_and_color) for one chunk of R rays.

    Returns the 12 tensors of ``render_rays_batch`` (scenerf.py:456-469) under the same keys,
    plus ``som_means`` and, if asked, stage intermediates used by the per-kernel parity tests.

    ``head_offsets`` (R, G, 2), parity tests only: the VALUE of the gaussian head's output is replaced by this tensor while its
    gradient still flows through the head computed here (straight-through).  With the offsets a reduced-precision
    implementation produced, everything downstream -- sample positions, sort order, sphere indices -- is evaluated at that
    implementation's sample positions, so the two can be compared sample by sample instead of through the chaos of samples
    that cross a texel boundary.  None (the default) is the reference's arithmetic, unchanged.

    ``sphere_idx = (main (R*N, 2), head (R*G, 2))``, parity tests only: the features are gathered at these integer sphere pixels
    instead of the ones computed here (which are still returned as ``_idx`` / ``_idx_g``).  The indices go through acos / atan2,
    whose last ulp differs between libm implementations; a test first checks that the other implementation's indices equal the
    ones computed here except within rounding noise of a .5 boundary, then evaluates the oracle AT those indices so that the
    remaining comparison is arithmetic only.

    ``som_choices = (bmu (R, N), mask (R, G))``, parity tests only: RaySOM's two discrete choices, see ``ray_som_kl``.
    """
    inv_K = torch.inverse(cam_K)
    R = pixels.shape[0]
    G, P, D = cfg.n_gaussians, cfg.n_pts_per_gaussian, cfg.max_sample_depth
    # uniform samples are drawn in every configuration (scenerf.py:609-616): n_pts_uni of them, or scenerf_bf.py:623-626's substitute of 2
    # when that is 0; the KITTI model has no substitute and divides by zero in uniform_sampling (utils.py:77)
    U = cfg.n_pts_uni if cfg.n_pts_uni > 0 else cfg.uni_fallback
    if U == 0 and P == 1:
        raise ZeroDivisionError("float division by zero (utils.py:77: step = (d_max - d_min) / n_pts_per_ray with n_pts_uni == 0)")
    # (U == 0 with P > 1: the KITTI reference divides by zero here as well; the product renders the gaussian samples alone -- a superset
    # of the reference -- and this restatement follows it so that the configuration can be checked: no uniform samples are drawn)
    rule = cfg.index_rule
    dirs, unit = ray_directions(pixels, inv_K, rule)
    viewdir = _matvec(T_source2infer[:3, :3], dirs, rule)                          # utils.py:170

    # uniform samples (utils.py:112-173); with n_pts_uni == 0 and P > 1 they are drawn and not rendered (noise_u may then be empty)
    if noise_u.shape[1] == 0 and cfg.n_pts_uni == 0 and P != 1:
        noise_u = torch.zeros(R, U, 1, device=pixels.device)
    if U > 0:
        dist_u = uniform_distances(R, U, D, noise_u)
        pts_u_src = dist_u.unsqueeze(-1) * unit.reshape(R, 1, 3)
        z_u = pts_u_src[:, :, 2]
        pts_u = to_frame(pts_u_src.reshape(-1, 3), T_source2infer, rule).reshape(R, U, 3)
    else:
        dist_u = z_u = torch.zeros(R, 0, device=pixels.device)
        pts_u = torch.zeros(R, 0, 3, device=pixels.device)

    # gaussian heads (scenerf.py:549-596)
    anchors = gaussian_anchor_distances(cfg).type_as(cam_K).reshape(1, G, 1).expand(R, -1, 1)
    apts = to_frame((anchors * unit.reshape(R, 1, 3)).reshape(-1, 3), T_source2infer, rule)
    vd_g = viewdir.unsqueeze(1).expand(-1, G, -1).reshape(-1, 3)
    xin_g, idx_g = point_inputs(apts, vd_g, x_rgb, cam_K, cfg, None if sphere_idx is None else sphere_idx[1])
    keep_g = {} if keep_intermediates else None
    off = resnetfc_forward(mlp_gaussian, xin_g, keep=keep_g).reshape(R, G, 2)
    off_own = off
    if head_offsets is not None:
        off = off + (head_offsets.reshape(R, G, 2).to(off.dtype) - off).detach()
    g_means = torch.relu(anchors.squeeze(-1) + off[:, :, 0]) + cfg.gauss_floor
    g_stds = torch.relu(off[:, :, 1] + cfg.std) + cfg.gauss_floor

    # gaussian samples (utils.py:186-229)
    dist_g = g_means.repeat_interleave(P, dim=1) + noise_g * g_stds.repeat_interleave(P, dim=1)
    dist_g = torch.where(dist_g < 0.1, torch.full_like(dist_g, 0.1), dist_g)
    pts_g_src = dist_g.unsqueeze(-1) * unit.reshape(R, 1, 3)
    z_g = pts_g_src[:, :, 2]
    pts_g = to_frame(pts_g_src.reshape(-1, 3), T_source2infer, rule).reshape(R, G * P, 3)

    # merge + sort (scenerf.py:636-659)
    if cfg.n_pts_uni > 0:
        pts = torch.cat([pts_u, pts_g], dim=1)
        zz = torch.cat([z_u, z_g], dim=1)
        dist = torch.cat([dist_u, dist_g], dim=1)
    elif P == 1:
        pts, zz, dist = pts_u, z_u, dist_u
    else:
        pts, zz, dist = pts_g, z_g, dist_g
    perm = torch.argsort(dist, dim=1)
    dist = torch.gather(dist, 1, perm)
    zz = torch.gather(zz, 1, perm)
    pts = torch.gather(pts, 1, perm.unsqueeze(-1).expand(-1, -1, 3))
    N = dist.shape[1]

    # main MLP on detached points (scenerf.py:661-665, 505-541)
    vd = viewdir.unsqueeze(1).expand(-1, N, -1).reshape(-1, 3)
    xin, idx = point_inputs(pts.detach().reshape(-1, 3), vd, x_rgb, cam_K, cfg, None if sphere_idx is None else sphere_idx[0])
    keep_m = {} if keep_intermediates else None
    out = resnetfc_forward(mlp, xin, keep=keep_m)
    color_s = torch.sigmoid(out[..., :3]).reshape(R, N, 3)
    density = density_activation(out[..., 3:4]).reshape(R, N)

    comp = composite(density, dist, zz, color_s)
    som_info = {} if keep_intermediates else None
    loss_kl, som_means, som_vars, bmu = ray_som_kl(g_means, g_stds, dist, comp["alphas"], cfg.som_sigma,
                                                   cfg.kl_std_floor, bmu_use=None if som_choices is None else som_choices[0],
                                                   mask_use=None if som_choices is None else som_choices[1], info=som_info)
    ret = {
        "depth": comp["depth"], "color": comp["color"], "gaussian_means": g_means, "gaussian_stds": g_stds,
        "weights_at_depth": comp["weights_at_depth"], "closest_pts_to_depths": comp["closest"],
        "loss_kl": loss_kl, "alphas": comp["alphas"], "som_vars": som_vars, "densities": density,
        "weights": comp["weights"], "depth_volumes": zz, "som_means": som_means,
    }
    if keep_intermediates:
        ret.update({
            "_dirs": dirs, "_unit": unit, "_viewdir": viewdir, "_dist_u": dist_u, "_anchor_pts": apts,
            "_idx_g": idx_g, "_xin_g": xin_g, "_offsets": off, "_offsets_own": off_own, "_dist_g": dist_g, "_perm": perm,
            "_dist_sorted": dist, "_pts_sorted": pts, "_idx": idx, "_xin": xin, "_mlp_out": out,
            "_colors": color_s, "_closest_idx": comp["closest_idx"], "_bmu": bmu,
            "_keep_mlp": keep_m, "_keep_gauss": keep_g, "_som_info": som_info,
        })
    return ret


def render_rays_batch(cfg, mlp, mlp_gaussian, cam_K, T_source2infer, x_rgb, sampled_pixels, noise_u, noise_g,
                      ray_batch_size: int = 128):
    """scenerf.py:392-471: chunk loop + concatenation (noise is sliced per chunk)."""
    outs = []
    for s in range(0, sampled_pixels.shape[0], ray_batch_size):
        e = s + ray_batch_size
        outs.append(render_chunk(cfg, mlp, mlp_gaussian, cam_K, T_source2infer, x_rgb,
                                 sampled_pixels[s:e], noise_u[s:e], noise_g[s:e]))
    return {k: torch.cat([o[k] for o in outs], dim=0) for k in outs[0]}


def training_proxy_loss(ret: Dict[str, torch.Tensor]) -> torch.Tensor:
    """Scalar touching every gradient edge the reference losses use (scenerf.py:290-307)."""
    return ret["depth"].mean() + ret["color"].mean() + ret["loss_kl"].mean() + ret["gaussian_means"].mean()
