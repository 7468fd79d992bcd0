"""Direct parity of the TIMED path against the CPU oracle at the sizes bench.py and BASELINE.json name.

What bench.py times is: bf16 operands, the fused forward trunk and dgrad chain (wide.hip: 128-row blocks), the batched transposing-read
weight-gradient launch (wgrad.hip, >= 32768 rows) and the feature-gradient kernel (dfeat.hip), at KITTI 1500x452, R = 1200
rays x N = 128 samples in ONE chunk.  The reference-minted golden vectors run R <= 64 rays; this file closes the gap: the very
configurations of BASELINE.json configs[1] (KITTI, as benched), configs[3] (BundleFusion, R = 1080 x N = 96) and one
configs[4]-sized chunk (N = 512, 16,384 rows) are rendered + back-propagated on the GPU and compared, output by output and
gradient by gradient (both MLPs' 20 tensors each, the 5 feature maps), with `oracle.render_chunk` + torch autograd on the same
pixels, noise, weights and WHITE-NOISE maps (a +-1 sphere index picks an unrelated texel: nothing hides an index error).

What is discrete on this path, and how each is held:
  * spherical indices (bit-exact).  Round 5: the whole geometry chain -- ray direction, sample point, projected pixel, both
    angles, round() -- follows torch-CPU's operation sequence (csrc/sphere_exact.h, held bit for bit on the CPU by
    tests/test_sphere_exact.py).  acos is pinned to SLEEF u10 because torch.acos is not one function (MKL's vmsAcos, whose last bit
    depends on the host's instruction set; oracle/sleef_acos.py).  Step 1 requires EVERY index of the GPU to equal the oracle's under that
    rule at identical head offsets -- no teacher-forced indices, no window around the .5 boundaries -- and reports how many rows the
    reference as this host runs it would place on the neighbouring texel (columns: never).  Independently, EVERY index must be a correct
    rounding of the float64 evaluation of spherical_mapping.py:99-115 on the same fp32 point up to a window derived from the fp32 unit
    roundoff of each stage (``_sphere_f64``).
  * the gaussian head's offsets are MLP outputs: they carry MFMA accumulation order (fp32, ~1e-6 relative) or bf16 rounding (~1e-2 m),
    which moves the 4*P gaussian samples of every ray; on white-noise maps a sample that crosses a texel boundary reads unrelated features.
Step 2 therefore evaluates the oracle AT the GPU's head offsets (`render_chunk(head_offsets=)`: values substituted, gradients still
through the oracle's own head) and requires EVERY ray -- fraction 1.0, no exemptions -- to meet the per-ray gates and every gradient
tensor its relative-L2 gate: that comparison is arithmetic only (fp32: MFMA accumulation order; bf16: the fused forward, the fused dgrad
chain, the batched weight gradients, the feature scatter).  The head itself is compared directly (GPU offsets vs the offsets the oracle's
head computes from the same indices).  Step 3 reports the free-running comparison against the unmodified oracle (``index_rule="torch"``),
texel-crossing chaos included, and gates its summary statistics.

loss_kl / som_vars: RaySOM makes two more discrete choices.  Its BMU is an argmax over values that tie at the additive floors (1e-5,
1e-8) for samples far from every gaussian (ray_som_kl.py:46-52; in the KITTI golden every ray has samples with a relative margin
< 1e-5 and those samples carry O(1) weight), and its update mask thresholds |d mean|, |d std| at 0.1 (:66-70).  They are treated
like the sphere indices: step 1b compares the GPU's BMU (per sample) and mask (per gaussian) with the oracle's own and requires
every difference to sit on a tie (argmax margin / threshold distance below the arithmetic noise of the precision, ``SOM_TIE``);
step 2 then evaluates the oracle AT the GPU's choices (``som_choices``), so that loss_kl is gated on EVERY ray and the gradient
gates of the gaussian head and the maps are SURVEY 8d's 1e-3 in fp32, like the radiance MLP's.  Step 3 compares with the unmodified
oracle (free-running) in both precisions: in fp32 every ray none of whose discrete choices differs must meet the 8d gates, and
every gradient must be within 1e-3 plus the (oracle-vs-oracle) effect of the differing choices.

Gates: fp32 = SURVEY section 8d (depth rel 1e-4, colour abs 1e-5, grads rel 1e-3); the rest = values measured on MI355X (printed
by this test and stored in gpurun_out/parity_full_*.json when that directory exists) times two."""
import json
import os

import pytest
import torch

import scenerf_oracle as orc
from golden_util import OUT_KEYS
from scenerf_amd import synth
from scenerf_amd.model import SceneRF, SceneRFBundleFusion
from scenerf_amd.renderer import MLP_PARAM_NAMES

pytestmark = pytest.mark.gpu
DEV = "cuda"

CASES = {
    # BASELINE.json configs[1] exactly as bench.py runs it (make_model / step): one chunk of 1200 rays, U=64, G=4, P=16
    "kitti_c2_r1200_n128": dict(variant="kitti", R=1200, U=64, P=16, sphere=(1500, 452), img=(1220, 370), pose=(1.0, 0.0), seed=900),
    # BASELINE.json configs[3]: BundleFusion 640x480, sphere 960x720 (CLI), R = 1080 (train_bundlefusion.py:32), N = 96
    "bf_c4_r1080_n96": dict(variant="bf", R=1080, U=64, P=8, sphere=(960, 720), img=(640, 480), pose=(0.3, 8.0), seed=910),
    # one chunk at BASELINE.json configs[4]'s sampling: N = 512 (U=256, P=64); 32 rays = 16,384 rows (fused kernels)
    "kitti_c5_r32_n512": dict(variant="kitti", R=32, U=256, P=64, sphere=(1500, 452), img=(1220, 370), pose=(2.0, 5.0), seed=920),
    # the reference's own CLI default (train_kitti.py:33-35: n_pts_uni = 32, n_pts_per_gaussian = 8 -> N = 64) at its n_rays = 1200: 76,800
    # rows = 600 row blocks of the 128-row kernels (two rays per block), the C = 1 instantiations of the per-ray tail
    "kitti_default_r1200_n64": dict(variant="kitti", R=1200, U=32, P=8, sphere=(1500, 452), img=(1220, 370), pose=(1.0, 0.0), seed=930),
}

# ---- regression gate: the values MEASURED on MI355X for the committed kernels (tests/golden/parity_full_measured.json, written by
# tools/make_parity_reference.py from a run's gpurun_out/parity_full_*.json) -- a result may not be more than REGRESSION x worse than
# what was measured (with small floors: values at rounding noise move from run to run through the fp32 atomics' order).  The absolute
# gates below stay as the outer bound; this one makes a 1.9 x degradation inside them visible.
REGRESSION = 1.25
FREE_REGRESSION = 2.0        # free-running bf16 gradients against the oracle at its own discrete choices (noisier: choices flip)
_MEASURED_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parity_full_measured.json")
MEASURED = json.load(open(_MEASURED_PATH)) if os.path.exists(_MEASURED_PATH) else {}


def _regression_fails(key, rep):
    ref = MEASURED.get(key)
    if not ref:
        return []
    fails = []
    def chk(what, got, want, floor):
        if got > max(REGRESSION * want, floor):
            fails.append("[regression] %s: %.3e against %.3e measured for the committed kernels (x %.2f > %.2f)" % (what, got, want, got / max(want, 1e-300), REGRESSION))
    for k, v in ref.get("out", {}).items():
        m = rep["matched"]["out"].get(k)
        if m is not None:
            chk("output " + k, m["max_abs"] if k in ABS_KEYS else m["max_rel"], v, 2e-6)
    groups = {}
    for nm, v in rep["matched"]["grad"].items():
        if "rel_l2" in v:
            g = nm.split(".")[0] + "."
            groups[g] = max(groups.get(g, 0.0), v["rel_l2"])
    for g, v in ref.get("grad", {}).items():
        if g in groups:
            chk("gradients " + g + "* (worst tensor, rel L2)", groups[g], v, 2e-4)
    if "loss" in ref:
        chk("proxy loss rel", rep["matched"]["loss"]["rel"], ref["loss"], 2e-6)
    if "head" in ref:
        chk("gaussian head offsets rel L2", rep["head_offsets"]["rel_l2"], ref["head"], 1e-6)
    # free-running bf16 gradients (round 6): 2 x the case's own measured worst tensor per group instead of one 0.2 / 0.4 / 0.4 for all
    # cases ("same direction") -- GRAD_GATE["free"] stays as the outer bound for a case without a measured entry
    if "free_grad" in ref and "free" in rep and "grad" in rep["free"]:
        fg = {}
        for nm, v in rep["free"]["grad"].items():
            if "rel_l2" in v:
                g = nm.split(".")[0] + "."
                fg[g] = max(fg.get(g, 0.0), v["rel_l2"])
        for g, v in ref["free_grad"].items():
            if g in fg and fg[g] > max(FREE_REGRESSION * v, 2e-3):
                fails.append("[regression] free-running gradients %s* (worst tensor, rel L2): %.3e against %.3e measured (x %.2f > %.2f)" % (
                    g, fg[g], v, fg[g] / max(v, 1e-300), FREE_REGRESSION))
    return fails

# ---- gates -------------------------------------------------------------------------------------------------------------------
# per-ray: |got - ref| <= tol * (1 + |ref|) ("rel" keys) or <= tol (ABS_KEYS), required of EVERY ray in the matched comparison.
ABS_KEYS = ("color", "alphas", "weights")
OUT_GATE = {
    # depth rel 1e-4, colour abs 1e-5; the (R, N) outputs at 2 x measured (alphas 3.4e-5, weights 2.9e-6 rel, densities 2.3e-5)
    "fp32": dict(depth=1e-4, color=1e-5, gaussian_means=1e-6, gaussian_stds=1e-6, depth_volumes=1e-6, alphas=1e-4, weights=2e-5, densities=1e-4),
    # measured at identical sample positions (KITTI R=1200 worst case): depth 3.9e-4, colour 1.5e-4, alphas 1.6e-3, weights 1.1e-4, densities 8e-4
    # at N = 128 and 2.04e-3 on one ray of 1,200 at N = 64 (these are the OUTER bounds: each case is also held to 1.25 x its own measured
    # values, REGRESSION above)
    "bf16": dict(depth=1e-3, color=3e-4, gaussian_means=1e-6, gaussian_stds=1e-6, depth_volumes=1e-6, alphas=3e-3, weights=2.5e-4, densities=3e-3),
}


def _out_gate(precision, N):
    """alpha = 1 - exp(-sigma delta) carries a density error through the sample spacing delta: its absolute gate, set at N = 128,
    scales with 128 / N below that (measured at the reference's default N = 64: bf16 3.4e-3 against 1.6e-3 at N = 128)."""
    g = dict(OUT_GATE[precision])
    g["alphas"] *= max(1.0, 128.0 / N)
    return g
# relative L2 of a whole gradient tensor, by group.  fp32: SURVEY 8d's 1e-3 for the radiance MLP (measured <= 3.7e-4); the gaussian
# head and the maps also carry the KL term's gradient, and ~2 % of the rays flip a RaySOM mask term (docstring): measured 1.3e-3 /
# 1.5e-3 at KITTI R = 1200.  bf16 (identical positions): measured 2.4e-2 / 5.6e-2 / 6.7e-2 (worst case: the N = 512 chunk).
GRAD_GATE = {"fp32": {"mlp.": 1e-3, "mlp_gaussian.": 1e-3, "x_rgb.": 1e-3},
             "bf16": {"mlp.": 5e-2, "mlp_gaussian.": 1.2e-1, "x_rgb.": 1.4e-1},
             "free": {"mlp.": 2e-1, "mlp_gaussian.": 4e-1, "x_rgb.": 4e-1}}
LOSS_GATE = {"fp32": 2e-5, "bf16": 5e-4, "free": 1e-3}         # relative error of the training proxy loss
HEAD_GATE = {"fp32": 2e-5, "bf16": 1.5e-2}                     # relative L2 of the gaussian head's offsets (measured 6.4e-3 in bf16)
# loss_kl (docstring: BMU ties): error of the mean over the chunk's rays (one flipped ray of 32 moves it by percents) and the fraction
# of rays within 2e-4 (fp32) / 2e-2 (bf16).  Measured: fp32 1.5e-3 / 0.981 (R = 1200); bf16 1.8e-3 / 0.964 (R = 1200), 5.4e-2 / 0.9375 (R = 32)
# with RaySOM's discrete choices matched (step 2) loss_kl is gated per ray, every ray: |got - ref| <= tol (1 + |ref|)
# measured (profiles/r03_f_parity_full_*.json): fp32 max rel 1.3e-5, mean rel 2.6e-7; bf16 max rel 9.7e-5, mean rel 2.0e-5
KL_GATE = {"fp32": dict(tol=5e-5, mean_rel=2e-6), "bf16": dict(tol=5e-4, mean_rel=1e-4)}
# a differing BMU / mask entry must sit on a tie: relative argmax margin / distance from the 0.1 threshold below this
# measured: the BMU differs on 0-93 of 153,600 samples, every one at a relative argmax margin <= 1.2e-7 (one fp32 ulp: exact ties at the
# additive floors), in both precisions; the mask never differs
SOM_TIE = {"fp32": dict(bmu=1e-6, mask=1e-4, max_frac_bmu=2e-3, max_frac_mask=1e-3), "bf16": dict(bmu=1e-6, mask=1e-3, max_frac_bmu=4e-3, max_frac_mask=2e-3)}
# free-running fp32 (step 3), rays without a differing discrete choice: SURVEY 8d for depth / colour; the head's own outputs carry its fp32
# MFMA rounding (offsets rel L2 <= 2e-5 of up to 100 m) and loss_kl amplifies it -- measured max rel: means 2.7e-6, stds 3.7e-5,
# depth_volumes 1.1e-5, loss_kl 5.9e-4
FREE_FP32_GATE = dict(depth=1e-4, color=1e-5, alphas=1e-4, weights=2e-5, densities=1e-4, gaussian_means=1e-5, gaussian_stds=1e-4,
                      depth_volumes=5e-5, loss_kl=2e-3)
# alpha = 1 - exp(-sigma delta): the same density error (gated relative, above) seen through the sample spacing.  The 1e-4 was set at
# N = 128 (measured 3.2e-5); the reference's default N = 64 doubles the spacing (measured 8.6e-5 (1 + alpha), densities 4.3e-5 as at
# N = 128): the absolute gate scales with 128 / N below 128
def _free_alpha_gate(N):
    return FREE_FP32_GATE["alphas"] * max(1.0, 128.0 / N)
FREE_FP32_MAX_TOUCHED_RAYS = 0.10                              # rays with any differing discrete choice (index, order, BMU, mask) vs the free oracle
FREE_BF16_GATE = dict(depth_rel_median=1e-3, depth_rel_p99=5e-3, color_abs_p99=2e-3, gaussian_means_rel_max=3e-3)   # measured 3.0e-4 / 2.3e-3 / 6.6e-4 / 1.5e-3


def _ulp32(x):
    """Spacing of fp32 numbers at |x| (x: float64 tensor)."""
    return torch.pow(2.0, torch.floor(torch.log2(x.abs().clamp(min=1e-30))) - 23)


def _sphere_f64(pts32, K32, ocfg):
    """spherical_mapping.py:99-115 behind utils.py:298-315 in float64 on the fp32 points: (coordinate (M,2), window (M,2), valid (M,)).
    The window bounds how far an fp32 evaluation of the same chain may land from the float64 value, stage by stage in units of the
    fp32 spacing at each stage's magnitude (4 roundings per stage: products, sums and the libm call, which is where implementations
    differ): the point itself (the GPU's transform into the infer frame may round differently: 2 ulp of |p|, seen through f/z), the
    projected pixel (4 ulp at |pix|), the angle in degrees (pixel error through <= (180/pi)/f deg per px, + 4 ulp at 180), the sphere
    pixel (angle error x (W-1)/fov, + 2 ulp at W)."""
    p, K = pts32.double(), K32.double()
    h = (K @ p.T).T
    valid = h[:, 2] > 0
    z = h[:, 2].clamp(min=1e-9)
    pix = h[:, :2] / z[:, None]
    f = float(min(K[0, 0], K[1, 1]))
    e_pix = 4 * _ulp32(pix.abs().clamp(min=1.0)) + (2 * _ulp32(p.abs().max(dim=1).values) * f / z)[:, None]
    c = (torch.inverse(K) @ torch.cat([pix, torch.ones_like(pix[:, :1])], 1).T).T
    n = c.norm(dim=1)
    v_min, v_fov, h_min, h_fov = ocfg.fov
    import math
    v = torch.acos(-c[:, 1] / n) / math.pi * 180
    hh = 180 - torch.atan2(c[:, 2], c[:, 0]) / math.pi * 180
    x = torch.stack([(hh - h_min) / h_fov * (ocfg.sphere_W - 1), (v - v_min) / v_fov * (ocfg.sphere_H - 1)], 1)
    u180 = float(_ulp32(torch.tensor(180.0, dtype=torch.float64)))
    e_ang = e_pix.max(dim=1).values * (180 / math.pi) / f + 4 * u180
    scale = torch.tensor([(ocfg.sphere_W - 1) / h_fov, (ocfg.sphere_H - 1) / v_fov], dtype=torch.float64)
    w = e_ang[:, None] * scale[None] + 2 * _ulp32(x.abs().clamp(min=1.0))
    return x, w, valid


def _inputs(spec):
    sd = spec["seed"]
    W, H = spec["sphere"]
    mlp, mlpg = synth.mlp_state(sd + 1, 4), synth.mlp_state(sd + 2, 2, out_scale=4.0 if spec["variant"] == "kitti" else 0.5)
    maps = synth.feature_maps(W, H, sd + 3, smooth=False)
    pix = synth.stride2_pixels(spec["img"], spec["R"], sd + 4)
    nu, ng = synth.sampling_noise(spec["R"], spec["U"], 4 * spec["P"], sd + 5)
    K = synth.kitti_cam_K() if spec["variant"] == "kitti" else synth.bundlefusion_cam_K()
    T = synth.rel_pose(*spec["pose"])
    return mlp, mlpg, maps, pix, nu, ng, K, T


def _ctor(spec):
    if spec["variant"] == "kitti":
        return SceneRF, dict(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8, n_pts_uni=spec["U"], n_pts_per_gaussian=spec["P"])
    return SceneRFBundleFusion, dict(som_sigma=0.02, std=0.1, add_fov_hor=14, add_fov_ver=11, sphere_W=spec["sphere"][0],
                                     sphere_H=spec["sphere"][1], n_pts_uni=spec["U"], n_pts_per_gaussian=spec["P"], max_sample_depth=12)


def _oracle_run(name, head_offsets=None, sphere_idx=None, som_choices=None, index_rule="torch"):
    """oracle.render_chunk + autograd of the proxy loss -> outputs, gradients, indices.  ``index_rule``: "torch" = the reference's call
    as this host runs it (the free-running comparison), "pinned" = the pinned rule (OracleConfig.index_rule; the matched comparison,
    where EVERY sphere index must then equal the GPU's)."""
    spec = CASES[name]
    mlp, mlpg, maps, pix, nu, ng, K, T = _inputs(spec)
    mk = orc.OracleConfig.kitti if spec["variant"] == "kitti" else orc.OracleConfig.bundlefusion
    ocfg = mk(n_pts_uni=spec["U"], n_pts_per_gaussian=spec["P"], index_rule=index_rule)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    po = {k: v.clone().requires_grad_(True) for k, v in mlp.items()}
    pg = {k: v.clone().requires_grad_(True) for k, v in mlpg.items()}
    xm = {k: v.clone().requires_grad_(True) for k, v in maps.items()}
    ref = orc.render_chunk(ocfg, po, pg, K, T, xm, pix, nu, ng, keep_intermediates=True, head_offsets=head_offsets, sphere_idx=sphere_idx,
                           som_choices=som_choices)
    loss = orc.training_proxy_loss(ref)
    loss.backward()
    grads = {"mlp." + n: po[n].grad for n in MLP_PARAM_NAMES}
    grads.update({"mlp_gaussian." + n: pg[n].grad for n in MLP_PARAM_NAMES})
    grads.update({"x_rgb." + k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in xm.items()})
    import dataclasses
    host, f64 = {}, {}   # the indices of the same points under torch.acos as THIS host runs it (MKL: statistics only), and in float64
    for key, pts in (("main", ref["_pts_sorted"].detach().reshape(-1, 3)), ("head", ref["_anchor_pts"].detach())):
        host[key] = orc.sphere_coords(orc.project_to_pixels(pts, K), torch.inverse(K), dataclasses.replace(ocfg, index_rule="torch"))
        f64[key] = _sphere_f64(pts, K, ocfg)
    si = ref["_som_info"]
    res = dict(out={k: ref[k].detach().clone() for k in OUT_KEYS}, loss=float(loss.item()), grads=grads, idx_host=host, f64=f64,
               som=dict(bmu=ref["_bmu"].clone(), mask=si["mask"].clone(), bmu_margin=si["bmu_margin"].clone(), mask_margin=si["mask_margin"].clone(),
                        means=ref["som_means"].detach().clone()),
               idx=dict(main=ref["_idx"].clone(), head=ref["_idx_g"].clone(), perm=ref["_perm"].clone(), closest=ref["_closest_idx"].clone()),
               dist_sorted=ref["_dist_sorted"].detach().clone(), offsets=ref["_offsets"].detach().clone(),
               offsets_own=ref["_offsets_own"].detach().clone())
    del ref
    return res


_FREE = {}


def _oracle_free(name):
    if name not in _FREE:
        _FREE[name] = _oracle_run(name)
    return _FREE[name]


def _within(got, ref, tol, absolute):
    got, ref = got.reshape(got.shape[0], -1), ref.reshape(ref.shape[0], -1)
    lim = tol if absolute else tol * (1.0 + ref.abs())
    return ((got - ref).abs() <= lim).all(dim=1)


def _compare(tag, o, out, grads, loss, R, rep, out_gate, grad_gate, loss_gate, cond=None):
    """Fill rep[tag] with measured errors of the 12 outputs, the proxy loss and every gradient; return the gate violations.
    ``cond`` {name: rel L2}: how far the ORACLE's own gradient moves when its sample positions move by one fp32 ulp (_run_case);
    added to the gate of that tensor and recorded next to the measured error."""
    r = rep[tag] = {"out": {}, "grad": {}}
    fails = []
    for k in OUT_KEYS:
        got, ref = out[k].detach().float().cpu(), o["out"][k]
        assert got.shape == ref.shape and bool(torch.isfinite(got).all()), k
        e, rr = (got - ref).reshape(R, -1), ref.reshape(R, -1)
        m = dict(rel_l2=float(e.double().norm() / max(float(rr.double().norm()), 1e-30)),
                 max_rel=float((e.abs() / (1.0 + rr.abs())).max()), max_abs=float(e.abs().max()))
        if out_gate is not None and k in out_gate:
            m["frac_rays_within_gate"] = float(_within(got, ref, out_gate[k], k in ABS_KEYS).float().mean())
            if m["frac_rays_within_gate"] < 1.0:
                fails.append("[%s] %s: %.4f of the rays within %.1e (max rel %.2e, max abs %.2e)" % (
                    tag, k, m["frac_rays_within_gate"], out_gate[k], m["max_rel"], m["max_abs"]))
        r["out"][k] = m
    r["loss"] = dict(got=float(loss), ref=o["loss"], rel=abs(float(loss) - o["loss"]) / abs(o["loss"]))
    if r["loss"]["rel"] > loss_gate:
        fails.append("[%s] proxy loss rel %.2e > %.1e" % (tag, r["loss"]["rel"], loss_gate))
    for nm, ref in o["grads"].items():
        g = grads[nm]
        rn = float(ref.double().norm())
        if rn == 0.0:   # pyramid levels no sample reaches (quirk Q1): exactly zero on both sides
            gz = 0.0 if g is None else float(g.abs().max())
            r["grad"][nm] = dict(ref_norm=0.0, got_max=gz)
            if gz != 0.0:
                fails.append("[%s] %s: non-zero (%.2e) where the oracle's gradient is exactly zero" % (tag, nm, gz))
            continue
        assert g is not None, nm
        gc = g.detach().double().cpu()
        rel = float((gc - ref.double()).norm() / rn)
        r["grad"][nm] = dict(rel_l2=rel, cosine=float((gc * ref.double()).sum() / (gc.norm() * rn)), ref_norm=rn)
        gate = [v for k, v in grad_gate.items() if nm.startswith(k)][0]
        if cond is not None:
            r["grad"][nm]["one_ulp_conditioning"] = cond.get(nm, 0.0)
            gate += cond.get(nm, 0.0)
        if rel > gate:
            fails.append("[%s] %s: gradient rel L2 %.2e > %.1e" % (tag, nm, rel, gate))
    worst = sorted(((v["rel_l2"], k) for k, v in r["grad"].items() if "rel_l2" in v), reverse=True)[:4]
    print("\n[%s] loss rel %.2e" % (tag, r["loss"]["rel"]))
    print("   outputs (max rel / max abs):", {k: "%.1e/%.1e" % (v["max_rel"], v["max_abs"]) for k, v in r["out"].items()})
    print("   worst gradients (rel L2):", ["%s %.2e" % (k, v) for v, k in worst])
    return fails


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("name", list(CASES))
def test_timed_path_matches_oracle_outputs_and_every_gradient(name, precision):
    _run_case(name, precision, "chw")


def test_timed_path_channels_last_entry_matches_oracle():
    """The configuration bench.py times by default: the feature maps arrive as (C,H,W) tensors with channels-last memory and are read in
    place (fp32 taps in the gather, gradients accumulated straight into the returned (H,W,C) memory) -- same oracle, same gates."""
    _run_case("kitti_c2_r1200_n128", "bf16", "hwc")


def _run_case(name, precision, entry):
    spec = CASES[name]
    mlp, mlpg, maps, pix, nu, ng, K, T = _inputs(spec)
    cls, kw = _ctor(spec)
    m = cls(precision=precision, **kw).to(DEV)
    m.mlp.load_state_dict(mlp)
    m.mlp_gaussian.load_state_dict(mlpg)
    m.debug_aux = True
    R, N = spec["R"], spec["U"] + 4 * spec["P"]
    if precision == "bf16":   # the path bench.py times: fused forward + fused dgrad chain (+ the batched wgrad launch from 32768 rows)
        assert m.render_cfg.uses_fused(R * N) and m.render_cfg.fused_backward and m.render_cfg.wgrad_tr
    if entry == "hwc":
        x = {k: torch.empty_strided(tuple(v.shape), (1, v.shape[2] * v.shape[0], v.shape[0]), device=DEV).copy_(v.to(DEV)).requires_grad_(True)
             for k, v in maps.items()}
    else:
        x = {k: v.to(DEV).requires_grad_(True) for k, v in maps.items()}
    out = m.render_rays_batch(K.to(DEV), T.to(DEV), x, sampled_pixels=pix.to(DEV), ray_batch_size=R, noise=(nu.to(DEV), ng.to(DEV)))
    loss = out["depth"].mean() + out["color"].mean() + out["loss_kl"].mean() + out["gaussian_means"].mean()
    loss.backward()
    torch.cuda.synchronize()
    aux = m.last_aux
    grads = {"mlp." + n: p.grad for n, p in zip(MLP_PARAM_NAMES, m.mlp.ordered_params())}
    grads.update({"mlp_gaussian." + n: p.grad for n, p in zip(MLP_PARAM_NAMES, m.mlp_gaussian.ordered_params())})
    grads.update({"x_rgb." + k: v.grad for k, v in x.items()})
    rep = {"case": name, "precision": precision, "rows": R * N, "maps": entry}
    fails = []

    # ---- the oracle at the GPU's head offsets and RaySOM choices (MLP arithmetic and argmax ties cannot be bit-exact; geometry is) ----
    off_gpu = aux["offsets"].detach().float().cpu().reshape(R, -1, 2)
    idx_main, idx_head = aux["sphere_idx"].cpu().long(), aux["sphere_idx_g"].cpu().long()
    bmu_gpu, mask_gpu = aux["bmu"].cpu().long(), aux["kl_mask"].detach().cpu() > 0.5
    o = _oracle_run(name, head_offsets=off_gpu, som_choices=(bmu_gpu, mask_gpu), index_rule="pinned")

    # step 1: indices -- SURVEY 8d "sphere indices bit-exact": at identical head offsets EVERY sphere index equals the oracle's under the
    # pinned rule (no teacher-forced indices, no window around the .5 boundaries); sorted distances / permutation bit-exact
    d_main, d_head = (idx_main != o["idx"]["main"]).any(dim=1), (idx_head != o["idx"]["head"]).any(dim=1)
    nflip, ntot = int(d_main.sum()) + int(d_head.sum()), d_main.numel() + d_head.numel()
    hm, hh = idx_main - o["idx_host"]["main"], idx_head - o["idx_host"]["head"]
    rep["index"] = dict(flipped_samples=nflip, samples=ntot, rays_touched=int((d_main.reshape(R, N).any(1) | d_head.reshape(R, -1).any(1)).sum()),
                        rows_differing_from_torch_on_this_host=int((hm[:, 1] != 0).sum()) + int((hh[:, 1] != 0).sum()),
                        columns_differing_from_torch_on_this_host=int((hm[:, 0] != 0).sum()) + int((hh[:, 0] != 0).sum()))
    if nflip:
        fails.append("%d of %d sphere indices differ from the oracle's under the pinned rule" % (nflip, ntot))
    if int(hm.abs().max()) > 1 or int(hh.abs().max()) > 1:    # (statistics otherwise: what THIS host makes of torch.acos and `K @ p`)
        fails.append("sphere indices vs the reference's calls on this host: off by more than one")
    # ... and against float64, independently of the oracle's fp32 arithmetic: EVERY index, the GPU's and the oracle's, is a rounding of
    # the float64 coordinate up to the ulp-derived window; a flipped sample therefore has both candidates adjacent to the float64 value
    f64rep = {}
    for key, ig, io, dflag in (("main", idx_main, o["idx"]["main"], d_main), ("head", idx_head, o["idx"]["head"], d_head)):
        x64, w64, valid = o["f64"][key]
        for who, ii in (("gpu", ig), ("oracle", io)):
            excess = ((ii.double() - x64).abs() - 0.5 - w64)[valid]
            f64rep["%s_%s_max_excess_px" % (key, who)] = float(excess.max()) if excess.numel() else 0.0
            if excess.numel() and float(excess.max()) > 0:
                fails.append("%s: %d %s sphere indices are not a rounding of the float64 coordinate within the fp32 window (worst excess %.2e px)" % (
                    key, int((excess > 0).sum()), who, float(excess.max())))
        if bool((~valid).any()) and not torch.equal(ig[~valid], io[~valid]):
            fails.append("%s: indices of points behind the camera differ" % key)
        f64rep[key + "_window_px_max"] = float(w64[valid].max()) if bool(valid.any()) else 0.0
        if bool(dflag.any()):   # distance of the float64 coordinate from the .5 boundary at the flipped samples, in windows
            dd = ((x64 - torch.floor(x64) - 0.5).abs() / w64)[dflag]
            f64rep[key + "_flipped_boundary_dist_in_windows_max"] = float(dd.min(dim=1).values.max())
    rep["index"]["float64"] = f64rep
    ds = o["dist_sorted"]
    uniq = torch.ones_like(ds, dtype=torch.bool)
    uniq[:, 1:] &= ds[:, 1:] != ds[:, :-1]
    uniq[:, :-1] &= ds[:, 1:] != ds[:, :-1]
    rep["index"]["dist_sorted_equal"] = bool(torch.equal(aux["dist_sorted"].cpu(), ds))
    rep["index"]["perm_equal"] = bool(torch.equal(aux["perm"].cpu().long()[uniq], o["idx"]["perm"][uniq]))
    rep["index"]["closest_idx_equal_frac"] = float((aux["closest_idx"].cpu().long() == o["idx"]["closest"]).float().mean())
    if not (rep["index"]["dist_sorted_equal"] and rep["index"]["perm_equal"]):
        fails.append("sorted sample distances / sort permutation are not bit-exact at identical head offsets")
    if rep["index"]["closest_idx_equal_frac"] < (0.999 if precision == "fp32" else 0.98):
        fails.append("closest-sample index equal on %.4f of the rays" % rep["index"]["closest_idx_equal_frac"])
    print("\n%s %s: %d of %d sphere indices differ from the pinned rule; %d rows / %d columns differ from the reference's calls as this host runs them; float64: %s" % (
        name, precision, nflip, ntot, rep["index"]["rows_differing_from_torch_on_this_host"], rep["index"]["columns_differing_from_torch_on_this_host"],
        {k: "%.2e" % v for k, v in f64rep.items()}))

    # step 1b: RaySOM's discrete choices -- the GPU's BMU per sample and mask per gaussian equal the oracle's own (same alphas up to
    # rounding; the mask at the same BMU) except on ties
    tie = SOM_TIE[precision]
    d_bmu = bmu_gpu != o["som"]["bmu"]
    d_msk = mask_gpu != o["som"]["mask"].bool()
    worst_b = float(o["som"]["bmu_margin"][d_bmu].max()) if bool(d_bmu.any()) else 0.0
    worst_m = float(o["som"]["mask_margin"][d_msk].max()) if bool(d_msk.any()) else 0.0
    rep["raysom_choices"] = dict(bmu_differs=int(d_bmu.sum()), samples=d_bmu.numel(), bmu_worst_margin=worst_b, rays_with_bmu_diff=int(d_bmu.any(1).sum()),
                                 mask_differs=int(d_msk.sum()), gaussians=d_msk.numel(), mask_worst_margin=worst_m,
                                 som_means_max_rel=float(((aux["som_means"].cpu() - o["som"]["means"]).abs() / (1 + o["som"]["means"].abs())).max()))
    print("   RaySOM choices: BMU differs on %d of %d samples (%d rays; worst argmax margin %.2e), mask on %d of %d (worst threshold distance %.2e); "
          "som_means at matched choices max rel %.2e" % (int(d_bmu.sum()), d_bmu.numel(), int(d_bmu.any(1).sum()), worst_b, int(d_msk.sum()),
                                                         d_msk.numel(), worst_m, rep["raysom_choices"]["som_means_max_rel"]))
    if worst_b > tie["bmu"]:
        fails.append("a BMU differs from the oracle's where its argmax margin is %.2e (> %.1e: not a tie)" % (worst_b, tie["bmu"]))
    if worst_m > tie["mask"]:
        fails.append("a RaySOM mask term differs where the thresholded quantity is %.2e from 0.1 (> %.1e)" % (worst_m, tie["mask"]))
    if int(d_bmu.sum()) > tie["max_frac_bmu"] * d_bmu.numel() or int(d_msk.sum()) > tie["max_frac_mask"] * d_msk.numel():
        fails.append("too many differing RaySOM choices: BMU %d / %d, mask %d / %d" % (int(d_bmu.sum()), d_bmu.numel(), int(d_msk.sum()), d_msk.numel()))

    # the gaussian head on its own (same indices, the oracle's own head arithmetic)
    e = off_gpu - o["offsets_own"]
    rep["head_offsets"] = dict(max_abs=float(e.abs().max()), scale=float(o["offsets_own"].abs().max()), rel_l2=float(e.norm() / o["offsets_own"].norm()))
    if rep["head_offsets"]["rel_l2"] > HEAD_GATE[precision]:
        fails.append("gaussian head offsets rel L2 %.2e > %.1e" % (rep["head_offsets"]["rel_l2"], HEAD_GATE[precision]))

    # step 2: every ray, every gradient, arithmetic only
    cond = None
    if precision == "fp32":
        # The conditioning of the gradients themselves: the positional encoding runs to |x| f ~ 1e4 rad, so a sample position that
        # differs in its last fp32 bit (the GPU forms origin + t * dir with fused multiply-adds, torch with separate roundings) moves
        # cos(x f) f -- the derivative every position gradient goes through -- by up to 1e4 * 6e-8.  Measured rather than argued: the
        # oracle again, its gaussian samples (half of the rows; the uniform ones stay, so this under-states it) nudged by ONE ulp of
        # the head's offsets in a seeded random direction, same discrete choices.  How far the oracle's own gradient moves is added to
        # SURVEY 8d's 1e-3 per tensor and recorded (``one_ulp_conditioning``).  At the reference's default N = 64 (oracle against
        # itself, CPU): x_rgb.1_2 4.6e-4, the radiance MLP's lin_z weights 3.9-4.2e-4, x_rgb.1_1 3.1e-4 (the one tensor measured above
        # 1e-3 on the GPU: 1.07e-3), every other tensor <= 1.6e-5 -- the tensors that multiply white-noise features, whose sums cancel.
        gsign = torch.Generator().manual_seed(spec["seed"] + 77)
        away = torch.where(torch.rand(off_gpu.shape, generator=gsign) < 0.5, torch.full_like(off_gpu, float("inf")), torch.full_like(off_gpu, float("-inf")))
        o2 = _oracle_run(name, head_offsets=torch.nextafter(off_gpu, away), sphere_idx=(idx_main, idx_head), som_choices=(bmu_gpu, mask_gpu),
                         index_rule="pinned")
        cond = {nm: float((o2["grads"][nm].double() - g.double()).norm() / max(float(g.double().norm()), 1e-300)) for nm, g in o["grads"].items()}
        del o2
    fails += _compare("matched", o, out, grads, loss.item(), R, rep, _out_gate(precision, N), GRAD_GATE[precision], LOSS_GATE[precision], cond)
    kl_got, kl_ref = out["loss_kl"].detach().cpu(), o["out"]["loss_kl"]
    klg = KL_GATE[precision]
    rep["loss_kl"] = dict(mean_rel=abs(float(kl_got.mean()) - float(kl_ref.mean())) / abs(float(kl_ref.mean())),
                          frac_within=float(_within(kl_got, kl_ref, klg["tol"], False).float().mean()),
                          max_rel=float(((kl_got - kl_ref).abs() / (1 + kl_ref.abs())).max()))
    print("   head offsets rel L2 %.2e; loss_kl at matched choices: mean rel %.2e, max rel %.2e, frac within %.4f; closest idx equal %.4f" % (
        rep["head_offsets"]["rel_l2"], rep["loss_kl"]["mean_rel"], rep["loss_kl"]["max_rel"], rep["loss_kl"]["frac_within"],
        rep["index"]["closest_idx_equal_frac"]))
    if rep["loss_kl"]["frac_within"] < 1.0 or rep["loss_kl"]["mean_rel"] > klg["mean_rel"]:
        fails.append("loss_kl at matched RaySOM choices: mean rel %.2e, %.4f of the rays within %.1e" % (
            rep["loss_kl"]["mean_rel"], rep["loss_kl"]["frac_within"], klg["tol"]))

    # step 3: free-running against the unmodified oracle
    free = _oracle_free(name)
    if precision == "fp32":
        # rays none of whose discrete choices differs from the free oracle's must meet SURVEY 8d as they are; the others are counted
        same_perm = (aux["perm"].cpu().long() == free["idx"]["perm"]).all(1)
        touched = ((idx_main != free["idx"]["main"]).any(1).reshape(R, N).any(1) | (idx_head != free["idx"]["head"]).any(1).reshape(R, -1).any(1)
                   | ~same_perm)
        touched_kl = touched | (bmu_gpu != free["som"]["bmu"]).any(1) | (mask_gpu != free["som"]["mask"].bool()).any(1)
        fr = rep["free_fp32"] = dict(rays_with_a_differing_index_or_order=int(touched.sum()), rays_with_any_differing_choice=int(touched_kl.sum()), rays=R, out={})
        if int(touched_kl.sum()) > FREE_FP32_MAX_TOUCHED_RAYS * R:
            fails.append("[free fp32] %d of %d rays have a differing discrete choice" % (int(touched_kl.sum()), R))
        for k, tol in FREE_FP32_GATE.items():
            tol = _free_alpha_gate(N) if k == "alphas" else tol
            keep = ~(touched_kl if k == "loss_kl" else touched)
            got, ref = out[k].detach().float().cpu()[keep], free["out"][k][keep]
            ok = _within(got, ref, tol, k in ABS_KEYS)
            fr["out"][k] = dict(rays=int(keep.sum()), frac_within=float(ok.float().mean()), max_rel=float(((got - ref).abs() / (1 + ref.abs())).max()))
            if not bool(ok.all()):
                fails.append("[free fp32] %s: %d of %d rays without a differing choice miss %.1e" % (k, int((~ok).sum()), int(keep.sum()), tol))
        # gradients: within 1e-3 of the free oracle's plus the effect of the differing choices, measured oracle-vs-oracle
        fr["grad"] = {}
        for nm, gf in free["grads"].items():
            rn = float(gf.double().norm())
            if rn == 0.0:
                continue
            gg = grads[nm].detach().double().cpu()
            choice = float((o["grads"][nm].double() - gf.double()).norm() / rn)
            rel = float((gg - gf.double()).norm() / rn)
            fr["grad"][nm] = dict(rel_l2=rel, choice_effect=choice)
            if rel > 1e-3 + 1.05 * choice:
                fails.append("[free fp32] %s: gradient rel L2 %.2e > 1e-3 + choice effect %.2e" % (nm, rel, choice))
        wg = sorted(((v["rel_l2"], v["choice_effect"], k) for k, v in fr["grad"].items()), reverse=True)[:3]
        print("   free-running fp32: %d rays with a differing index/order, %d with any differing choice; untouched rays max rel: %s; worst grads (rel, choice effect): %s" % (
            int(touched.sum()), int(touched_kl.sum()), {k: "%.1e" % v["max_rel"] for k, v in fr["out"].items()},
            ["%s %.1e/%.1e" % (k, a_, b_) for a_, b_, k in wg]))
    del o
    if precision == "bf16":
        fails += _compare("free", free, out, grads, loss.item(), R, rep, None, GRAD_GATE["free"], LOSS_GATE["free"])
        dref, dgot = free["out"]["depth"], out["depth"].detach().cpu()
        rel = (dgot - dref).abs() / dref.abs().clamp(min=1e-3)
        cerr = (out["color"].detach().cpu() - free["out"]["color"]).abs().reshape(-1)
        gm = (out["gaussian_means"].detach().cpu() - free["out"]["gaussian_means"]).abs() / free["out"]["gaussian_means"].abs()
        fb = dict(depth_rel_median=float(rel.median()), depth_rel_p99=float(rel.quantile(0.99)), color_abs_p99=float(cerr.quantile(0.99)),
                  gaussian_means_rel_max=float(gm.max()))
        rep["free"]["summary"] = fb
        print("   free-running bf16:", {k: "%.2e" % v for k, v in fb.items()})
        for k, v in fb.items():
            if v > FREE_BF16_GATE[k]:
                fails.append("free-running bf16 %s = %.2e > %.1e" % (k, v, FREE_BF16_GATE[k]))
    tag = "%s_%s%s" % (name, precision, "" if entry == "chw" else "_" + entry)
    fails += _regression_fails(tag, rep)
    if os.path.isdir("gpurun_out"):
        with open(os.path.join("gpurun_out", "parity_full_%s.json" % tag), "w") as f:
            json.dump(rep, f, indent=1)
    assert not fails, "\n".join(fails)
