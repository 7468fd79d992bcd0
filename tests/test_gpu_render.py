"""End-to-end parity of SceneRF.render_rays_batch (HIP path, through the C ABI) against the golden vectors
minted from the reference, and against the CPU oracle at a larger size, forward and backward."""
import json
import os

import pytest
import torch

import scenerf_oracle as orc
from golden_util import CASES, OUT_KEYS, Golden
from scenerf_amd.model import SceneRF, SceneRFBundleFusion
from scenerf_amd.renderer import MLP_PARAM_NAMES

pytestmark = pytest.mark.gpu
DEV = "cuda"

# per-ray gates: |got - ref| <= tol * (1 + |ref|), colour / alphas / weights absolute.  fp32 = SURVEY 8d (depth rel 1e-4, colour abs
# 1e-5; measured max 2.7e-6 / 6e-7 at R = 1200).  bf16 = SURVEY 8d (depth rel 2e-2, colour abs 2e-2): against a golden vector the bf16
# run is free-running -- its gaussian samples sit ~1e-2 m off the reference's and some cross a texel boundary (measured max
# 8e-3 / 1.6e-3 at R = 1200 on white-noise maps, tests/test_gpu_parity_full.py); the arithmetic-only bf16 gates live in that file.
TOL = {"fp32": dict(depth=1e-4, color=1e-5, other=1e-4), "bf16": dict(depth=2e-2, color=2e-2, other=6e-2)}
ABS_KEYS = ("color", "alphas", "weights")
# gradient digests (norm, the reference's top-256 entries): relative.  bf16 and small R: a texel-crossing sample is 1 of ~1000
GRAD_TOL = {"fp32": dict(norm=1e-2, topk=1e-2), "bf16": dict(norm=1.5e-1, topk=2e-1)}


def build_model(g: Golden, precision: str):
    cls = SceneRF if g.variant == "kitti" else SceneRFBundleFusion
    m = cls(precision=precision, **g.ctor).to(DEV)
    mlp, mlpg = g.mlp_states()
    m.mlp.load_state_dict(mlp)
    m.mlp_gaussian.load_state_dict(mlpg)
    return m


def run_model(m, g: Golden, maps, grad=True):
    x = {k: v.to(DEV).requires_grad_(grad) for k, v in maps.items()}
    out = m.render_rays_batch(g.cam_K.to(DEV), g.T.to(DEV), x, T_cam2velo=torch.eye(4, device=DEV),
                              sampled_pixels=g.pixels.to(DEV), ray_batch_size=g.chunk,
                              noise=(g.noise_u.to(DEV), g.noise_g.to(DEV)))
    return out, x


def frac_within(a, b, tol, absolute=False):
    a, b = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    lim = tol if absolute else tol * (1.0 + b.abs())
    ok = ((a - b).abs() <= lim).all(dim=1)
    return float(ok.float().mean()), ok


# rays per case whose sphere index differs from the free-running oracle's (the host's torch.acos / `K @ p` against the pinned rule, and
# gaussian samples that carry the head's fp32 arithmetic): MEASURED on MI355X per call site (tests/golden/clean_mask_measured.json,
# regenerated from the counts every run leaves in gpurun_out/clean_mask_counts.json) -- a run may mask at most measured + 1 rays
# (round 6; before that n_rays // 16 for every case, two orders of magnitude above what is measured)
_CLEAN_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clean_mask_measured.json")
CLEAN_MEASURED = json.load(open(_CLEAN_PATH)) if os.path.exists(_CLEAN_PATH) else {}
_CLEAN_SEEN = {}


def _clean_gate(key, flipped, n_rays):
    _CLEAN_SEEN[key] = {"flipped_rays": flipped, "rays": n_rays}
    if os.path.isdir("gpurun_out"):
        path = os.path.join("gpurun_out", "clean_mask_counts.json")
        try:
            cur = json.load(open(path)) if os.path.exists(path) else {}
        except Exception:
            cur = {}
        cur.update(_CLEAN_SEEN)
        json.dump(cur, open(path, "w"), indent=1, sort_keys=True)
    want = CLEAN_MEASURED.get(key)
    limit = want["flipped_rays"] + 1 if want is not None else max(1, n_rays // 256)
    assert flipped <= limit, "[%s] %d of %d rays carry a sphere index off the free-running oracle's; %s" % (
        key, flipped, n_rays, "measured %d (+1 allowed)" % want["flipped_rays"] if want is not None else "no measured entry: limit %d" % limit)


def clean_mask(aux, o, ocfg, K, n_rays, T, key=None):
    """Rays (the first ``n_rays`` of the render) whose every sample and anchor got the sphere index the FREE-RUNNING oracle ``o`` computed
    (the reference's calls as this host runs them) -- the rays a stored / free-running output vector can be compared on.

    Round 5: the geometry chain is bit-exact (csrc/sphere_exact.h), so before the mask is built EVERY index of the GPU is required to be the
    pinned rule (OracleConfig.index_rule) applied to the GPU's own sample distances and ray directions -- no window around the .5 boundaries.
    What is left to differ from the free-running oracle, and is masked: gaussian samples whose distance carries the gaussian head's fp32
    output (MLP arithmetic: accumulation order, not bit-exact), and what the host makes of torch.acos / `K @ p` (oracle/sleef_acos.py)."""
    import dataclasses
    n_main, n_head = o["_idx"].shape[0], o["_idx_g"].shape[0]
    iK = torch.inverse(K)
    rule = dataclasses.replace(ocfg, index_rule="pinned")
    got, got_g = aux["sphere_idx"].cpu().long()[:n_main], aux["sphere_idx_g"].cpu().long()[:n_head]
    unit = aux["unit_dir"].cpu()[:n_rays]
    G = n_head // n_rays
    anchors = orc.gaussian_anchor_distances(ocfg).reshape(1, G, 1)
    apts = orc.to_frame((anchors * unit.reshape(n_rays, 1, 3)).reshape(-1, 3), T, "pinned")
    idx_g_rule = orc.sphere_coords(orc.project_to_pixels(apts, K, "pinned"), iK, rule)
    assert torch.equal(got_g, idx_g_rule), "anchor sphere indices differ from the pinned rule (%d rows)" % int((got_g != idx_g_rule).any(1).sum())
    ds = aux["dist_sorted"].cpu()[:n_rays]
    pts = orc.to_frame((ds.unsqueeze(-1) * unit.reshape(n_rays, 1, 3)).reshape(-1, 3), T, "pinned")
    idx_rule = orc.sphere_coords(orc.project_to_pixels(pts, K, "pinned"), iK, rule)
    assert torch.equal(got.clamp(-10**9, 10**9), idx_rule.clamp(-10**9, 10**9)), \
        "%d sample sphere indices differ from the pinned rule at the GPU's own sample positions" % int((got != idx_rule).any(1).sum())
    dm, dh = got - o["_idx"], got_g - o["_idx_g"]
    assert int(dm.abs().max()) <= 1 and int(dh.abs().max()) <= 1
    flipped = (dm != 0).any(dim=1).reshape(n_rays, -1).any(dim=1) | (dh != 0).any(dim=1).reshape(n_rays, -1).any(dim=1)
    _clean_gate(key or "rays%d_rows%d" % (n_rays, n_main), int(flipped.sum()), n_rays)
    return ~flipped


def _clean_rays(m, g: Golden, R):
    """clean_mask for a golden case (single chunk): the oracle is pinned on the reference (test_oracle_golden.py)."""
    ocfg = (orc.OracleConfig.kitti if g.variant == "kitti" else orc.OracleConfig.bundlefusion)(**g.cfg_kwargs())
    mlp, mlpg = g.mlp_states()
    outs = [orc.render_chunk(ocfg, mlp, mlpg, g.cam_K, g.T, g.feature_maps(), g.pixels[s:s + g.chunk], g.noise_u[s:s + g.chunk],
                             g.noise_g[s:s + g.chunk], keep_intermediates=True) for s in range(0, R, g.chunk)]
    o = {k: torch.cat([c[k] for c in outs], dim=0) for k in ("_idx", "_idx_g")}
    return clean_mask(m.last_aux, o, ocfg, g.cam_K, R, g.T, key="golden_" + g.name)


def _loss_kl_at_the_gpus_choices(m, g: Golden, R):
    """loss_kl of the oracle (pinned on the reference: test_oracle_golden.py) evaluated AT the GPU's gaussian-head offsets and RaySOM
    choices (BMU per sample, mask per gaussian: render_chunk(head_offsets=, som_choices=)) under the pinned acos rule -- the sphere indices
    are NOT handed over: they must come out equal -- after checking that every differing RaySOM choice sits on a tie of the oracle's own
    (argmax margin / threshold distance)."""
    ocfg = (orc.OracleConfig.kitti if g.variant == "kitti" else orc.OracleConfig.bundlefusion)(index_rule="pinned", **g.cfg_kwargs())
    mlp, mlpg = g.mlp_states()
    aux = m.last_aux
    N, G = aux["bmu"].shape[1], aux["kl_mask"].shape[1]
    out = []
    for s in range(0, R, g.chunk):
        e = min(s + g.chunk, R)
        off = aux["offsets"].detach().float().cpu().reshape(R, G, 2)[s:e]
        idx = (aux["sphere_idx"].cpu().long().reshape(R, N, 2)[s:e].reshape(-1, 2), aux["sphere_idx_g"].cpu().long().reshape(R, G, 2)[s:e].reshape(-1, 2))
        bmu, msk = aux["bmu"].cpu().long()[s:e], aux["kl_mask"].detach().cpu()[s:e] > 0.5
        with torch.no_grad():
            o = orc.render_chunk(ocfg, mlp, mlpg, g.cam_K, g.T, g.feature_maps(), g.pixels[s:e], g.noise_u[s:e], g.noise_g[s:e],
                                 keep_intermediates=True, head_offsets=off, som_choices=(bmu, msk))
        assert torch.equal(o["_idx"], idx[0]) and torch.equal(o["_idx_g"], idx[1]), "sphere indices at the GPU's head offsets differ from the pinned rule"
        si = o["_som_info"]
        d_b, d_m = bmu != o["_bmu"], msk != si["mask"].bool()
        assert not bool(d_b.any()) or float(si["bmu_margin"][d_b].max()) <= 1e-6, "a BMU differs from the oracle's away from a tie"
        assert not bool(d_m.any()) or float(si["mask_margin"][d_m].max()) <= 1e-4, "a RaySOM mask term differs away from its threshold"
        out.append(o["loss_kl"])
    return torch.cat(out)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("name", CASES)
def test_render_matches_reference_golden(name, precision):
    g = Golden(name)
    m = build_model(g, precision)
    R = g.pixels.shape[0]
    m.debug_aux = precision == "fp32"   # (bf16: free-running, the gaussian samples sit elsewhere anyway)
    out, x = run_model(m, g, g.feature_maps())
    assert set(out) == set(OUT_KEYS)
    clean = _clean_rays(m, g, R) if m.debug_aux else torch.ones(R, dtype=torch.bool)
    tol = TOL[precision]
    report, worst = {}, {}
    for k in OUT_KEYS:
        ref = g.out(k)
        got = out[k].detach().float().cpu()
        assert got.shape == ref.shape, k
        t = tol["depth"] if k in ("depth", "depth_volumes", "gaussian_means", "gaussian_stds") else tol["color"] if k == "color" else tol["other"]
        _, ok = frac_within(got, ref, t, k in ABS_KEYS)
        report[k] = float(ok[clean].float().mean())
        e = (got - ref).reshape(R, -1)[clean]
        worst[k] = float((e.abs() / (1.0 if k in ABS_KEYS else 1.0 + ref.reshape(R, -1)[clean].abs())).max())
    print("\n%s %s: %d/%d rays with oracle-identical indices; max errors %s" % (
        name, precision, int(clean.sum()), R, {k: "%.1e" % v for k, v in worst.items()}))
    # every ray (with reference-identical sample indices) within the gate -- fraction 1.0
    for k in ("depth", "color", "weights", "alphas", "densities", "gaussian_means", "gaussian_stds", "depth_volumes"):
        assert report[k] == 1.0, "%s: only %.3f of the rays within tolerance (max error %.2e)" % (k, report[k], worst[k])
    # loss_kl: RaySOM's BMU is an argmax over values that tie at the additive floors for samples far from every gaussian
    # (ray_som_kl.py:46-52), so single rays flip against the stored vector under any rounding difference ...
    assert report["loss_kl"] >= 0.85, report["loss_kl"]
    if m.debug_aux:
        # ... which is why, in fp32, EVERY ray is held to the oracle evaluated at the GPU's own discrete choices, each differing
        # choice being checked to sit on a tie (the oracle's own choices reproduce the stored vector: tests/test_oracle_golden.py)
        kl_ref = _loss_kl_at_the_gpus_choices(m, g, R)
        kl_got = out["loss_kl"].detach().float().cpu()
        bad = ((kl_got - kl_ref).abs() > 5e-5 * (1.0 + kl_ref.abs()))
        assert not bool(bad.any()), "loss_kl at matched choices: %d of %d rays off (max rel %.2e)" % (
            int(bad.sum()), R, float(((kl_got - kl_ref).abs() / (1.0 + kl_ref.abs())).max()))
    # aggregate (the training loss proxy) must agree closely
    loss = out["depth"].mean() + out["color"].mean() + out["loss_kl"].mean() + out["gaussian_means"].mean()
    ref_loss = float(g.z["loss"])
    lrel = abs(loss.item() - ref_loss) / abs(ref_loss)
    assert lrel <= (5e-4 if precision == "fp32" else 5e-3), lrel
    # gradients: norm + the reference's top-|g| entries
    loss.backward()
    gt = GRAD_TOL[precision]
    tensors = {}
    for pn, p in zip(MLP_PARAM_NAMES, m.mlp.ordered_params()):
        tensors["mlp." + pn] = p.grad
    for pn, p in zip(MLP_PARAM_NAMES, m.mlp_gaussian.ordered_params()):
        tensors["mlp_gaussian." + pn] = p.grad
    for key, v in x.items():
        tensors["x_rgb." + key] = v.grad if v.grad is not None else torch.zeros_like(v)
    bad, wn, wt = [], 0.0, 0.0
    for nm, grad in tensors.items():
        assert grad is not None, nm
        d = g.grad_digest(nm)
        flat = grad.detach().float().cpu().reshape(-1)
        nrm = float(flat.double().norm())
        en = abs(nrm - d["norm"]) / max(d["norm"], 1e-30) if d["norm"] > 0 else nrm
        s = float(d["val"].abs().max())
        err = (flat[d["idx"]] - d["val"]).abs()
        # (bf16 on a tiny chunk -- the uniform-only case renders 64 samples -- is free-running: one ray whose RaySOM mask / BMU or
        # ReLU gate falls the other way moves the four texels of its taps by tens of per cent; there the 95th percentile of the
        # top-k errors is gated instead of their maximum, measured 2.7e-3 of 4.3e-3 on one such texel quadruple)
        worst_e = float(err.max()) if not (precision == "bf16" and R * g.noise_g.shape[1] < 1024) else float(err.quantile(0.95))
        et = worst_e / max(s, 1e-30) if s > 0 else float(flat[d["idx"]].abs().max())
        wn, wt = max(wn, en), max(wt, et)
        if en > gt["norm"] or et > gt["topk"]:
            bad.append((nm, "norm %.2e topk %.2e" % (en, et)))
    print("   loss rel %.2e, loss_kl frac %.3f, worst gradient norm error %.2e, worst top-k error %.2e" % (lrel, report["loss_kl"], wn, wt))
    assert not bad, bad


def test_map_gradients_zero_without_inrange_scales():
    """Q1: scales whose samples are all out of range get exactly zero gradient."""
    g = Golden("kitti_full_n64")
    m = build_model(g, "fp32")
    out, x = run_model(m, g, g.feature_maps())
    (out["depth"].mean() + out["color"].mean()).backward()
    assert float(x["1_8"].grad.abs().max()) == 0.0 and float(x["1_16"].grad.abs().max()) == 0.0
    assert float(x["1_1"].grad.abs().max()) > 0.0


def test_inference_no_grad_and_determinism():
    g = Golden("kitti_small_n64")
    m = build_model(g, "bf16").eval()
    with torch.no_grad():
        a, _ = run_model(m, g, g.feature_maps(), grad=False)
        b, _ = run_model(m, g, g.feature_maps(), grad=False)
    for k in OUT_KEYS:
        assert torch.equal(a[k], b[k]), k   # forward is deterministic (no atomics on the forward path)
    assert not a["depth"].requires_grad


def test_plumbing_config_c1_runs_through_the_product():
    """BASELINE.json configs[0] -- 4k rays x 64 samples through a 4-layer 128-wide ResnetFC (n_blocks = 1), forward only -- through the
    PRODUCT on the GPU: the same ray pipeline (ray setup, both samplers, sort, encode, gather, per-ray tail) with the MLPs evaluated by
    ``scenerf_hip_resnetfc_forward`` (one fp32 MFMA GEMM per nn.Linear: the fused kernels are built for 3 x 512), against the golden
    vector the reference's own render_rays_batch produced with its mlp / mlp_gaussian swapped for that shape (make_golden.py)."""
    from scenerf_amd.model import ResnetFC
    g = Golden("c1_plumbing_r4096_n64")
    assert g.meta["mlp"] == dict(n_blocks=1, d_hidden=128) and g.meta["R"] == 4096 and g.chunk == 1024
    m = SceneRF(precision="fp32", **g.ctor)
    m.mlp = ResnetFC(d_in=42, d_out=4, n_blocks=1, d_hidden=128)
    m.mlp_gaussian = ResnetFC(d_in=42, d_out=2, n_blocks=1, d_hidden=128)
    mlp, mlpg = g.mlp_states()
    m.mlp.load_state_dict(mlp)
    m.mlp_gaussian.load_state_dict(mlpg)
    m = m.to(DEV)
    R = g.pixels.shape[0]
    m.debug_aux = True
    with torch.no_grad():
        out, _ = run_model(m, g, g.feature_maps(), grad=False)
    assert set(out) == set(OUT_KEYS)
    clean = _clean_rays(m, g, R)        # (at most measured + 1 rays masked: tests/golden/clean_mask_measured.json)
    tol = TOL["fp32"]
    for k in OUT_KEYS:
        got = out[k].detach().float().cpu()
        assert bool(torch.isfinite(got).all()), k
        if k in ("loss_kl", "som_vars", "closest_pts_to_depths", "weights_at_depth"):
            continue   # argmax / argmin / threshold outputs: ties (test_render_matches_reference_golden gates them at matched choices)
        t = tol["depth"] if k in ("depth", "depth_volumes", "gaussian_means", "gaussian_stds") else tol["color"] if k == "color" else tol["other"]
        if ("out/" + k) in g.z.files:
            ref = g.out(k)
            assert got.shape == ref.shape, k
            _, ok = frac_within(got, ref, t, k in ABS_KEYS)
            assert bool(ok[clean].all()), "%s: %.4f of the clean rays within tolerance (max err %.2e)" % (
                k, float(ok[clean].float().mean()), float(((got - ref).abs() / (1.0 if k in ABS_KEYS else 1.0 + ref.abs())).reshape(R, -1)[clean].max()))
        else:       # the (R, N) outputs are stored as digests: the reference's 256 largest entries
            d = g.out_digest(k)
            N = got.shape[1]
            rows = d["idx"] // N
            keep = clean[rows]
            gv, rv = got.reshape(-1)[d["idx"]][keep], d["val"][keep]
            if k == "alphas":      # alpha = 1 - exp(-sigma delta): the absolute gate was set at N = 128 and scales with the sample spacing
                t = t * max(1.0, 128.0 / N)     # (tests/test_gpu_parity_full.py::_out_gate; measured here 1.16e-4 at N = 64)
            lim = t if k in ABS_KEYS else t * (1.0 + rv.abs())
            assert bool(((gv - rv).abs() <= lim).all()), "%s digest: max err %.2e" % (k, float((gv - rv).abs().max()))
    # ... a gradient CAN be asked of these shapes since round 6 (test_generic_resnetfc_trains_against_the_reference below), in fp32: the
    # bf16 kernels are built for the 3 x 512 trunk, and say so
    mb = SceneRF(precision="bf16", **g.ctor)
    mb.mlp = ResnetFC(d_in=42, d_out=4, n_blocks=1, d_hidden=128)
    mb.mlp_gaussian = ResnetFC(d_in=42, d_out=2, n_blocks=1, d_hidden=128)
    mb = mb.to(DEV)
    x = {k: v.to(DEV) for k, v in g.feature_maps().items()}
    with pytest.raises(RuntimeError, match="precision='fp32'"):
        mb.render_rays_batch(g.cam_K.to(DEV), g.T.to(DEV), x, sampled_pixels=g.pixels[:64].to(DEV), ray_batch_size=64,
                             noise=(g.noise_u[:64].to(DEV), g.noise_g[:64].to(DEV)))


@pytest.mark.parametrize("name", ["c1_train_r256_n64", "generic_train_2x64_r96"])
def test_generic_resnetfc_trains_against_the_reference(name):
    """resnetfc.py:67-164 is generic AND differentiable: BASELINE configs[0]'s 1 block x 128 (and a 2 x 64 net over two chunks) through the
    product WITH autograd -- `scenerf_hip_resnetfc_forward_train` / `_backward`, one fp32-MFMA GEMM per nn.Linear and per gradient -- against
    the golden vectors the reference's own render_rays_batch + backward produced with its two MLPs swapped for that shape
    (make_golden.py): all 12 outputs, and the norm + the reference's 256 largest entries of every parameter gradient and map gradient."""
    from scenerf_amd.model import ResnetFC
    g = Golden(name)
    mk = g.meta["mlp"]
    m = SceneRF(precision="fp32", **g.ctor)
    m.mlp = ResnetFC(d_in=42, d_out=4, **mk)
    m.mlp_gaussian = ResnetFC(d_in=42, d_out=2, **mk)
    mlp, mlpg = g.mlp_states()
    m.mlp.load_state_dict(mlp)
    m.mlp_gaussian.load_state_dict(mlpg)
    m = m.to(DEV)
    R = g.pixels.shape[0]
    m.debug_aux = True
    out, x = run_model(m, g, g.feature_maps())
    clean = _clean_rays(m, g, R)
    tol = TOL["fp32"]
    for k in ("depth", "color", "weights", "alphas", "densities", "gaussian_means", "gaussian_stds", "depth_volumes"):
        ref, got = g.out(k), out[k].detach().float().cpu()
        t = tol["depth"] if k in ("depth", "depth_volumes", "gaussian_means", "gaussian_stds") else tol["color"] if k == "color" else tol["other"]
        _, ok = frac_within(got, ref, t, k in ABS_KEYS)
        assert bool(ok[clean].all()), "%s: %.4f of the clean rays within tolerance" % (k, float(ok[clean].float().mean()))
    loss = out["depth"].mean() + out["color"].mean() + out["loss_kl"].mean() + out["gaussian_means"].mean()
    lrel = abs(loss.item() - float(g.z["loss"])) / abs(float(g.z["loss"]))
    assert lrel <= 5e-4, lrel
    loss.backward()
    names = [n for n, _ in m.mlp.named_parameters()]
    tensors = {}
    for mod_name, mod in (("mlp", m.mlp), ("mlp_gaussian", m.mlp_gaussian)):
        for pn, p in mod.named_parameters():
            tensors["%s.%s" % (mod_name, pn)] = p.grad
    for key, v in x.items():
        tensors["x_rgb." + key] = v.grad if v.grad is not None else torch.zeros_like(v)
    assert set(g.grad_names()) == set(tensors), (sorted(set(g.grad_names()) ^ set(tensors)), names)
    gt = GRAD_TOL["fp32"]
    bad, wn, wt = [], 0.0, 0.0
    for nm, grad in tensors.items():
        assert grad is not None, nm
        d = g.grad_digest(nm)
        flat = grad.detach().float().cpu().reshape(-1)
        nrm = float(flat.double().norm())
        en = abs(nrm - d["norm"]) / max(d["norm"], 1e-30) if d["norm"] > 0 else nrm
        s = float(d["val"].abs().max())
        et = float((flat[d["idx"]] - d["val"]).abs().max()) / max(s, 1e-30) if s > 0 else float(flat[d["idx"]].abs().max())
        wn, wt = max(wn, en), max(wt, et)
        if en > gt["norm"] or et > gt["topk"]:
            bad.append((nm, "norm %.2e topk %.2e" % (en, et)))
    print("\n%s: loss rel %.2e, %d/%d clean rays, worst gradient norm error %.2e, worst top-k error %.2e" % (name, lrel, int(clean.sum()), R, wn, wt))
    assert not bad, bad


def test_larger_chunk_against_oracle_bf16_and_fp32():
    """R=256 rays x N=128 (config-2 sampling) on the small sphere, vs the CPU oracle run here."""
    from scenerf_amd import synth
    kw = dict(sphere_W=376, sphere_H=114, n_pts_uni=64, n_pts_per_gaussian=16)
    ocfg = orc.OracleConfig.kitti(**kw)
    R = 256
    mlp, mlpg = synth.mlp_state(11, 4), synth.mlp_state(12, 2, out_scale=4.0)
    maps = synth.feature_maps(376, 114, 13, smooth=True)
    pix = synth.stride2_pixels((1220, 370), R, 14)
    nu, ng = synth.sampling_noise(R, 64, 64, 15)
    K, T = synth.kitti_cam_K(), synth.rel_pose(2.0, 10.0)
    ref = orc.render_chunk(ocfg, mlp, mlpg, K, T, maps, pix, nu, ng, keep_intermediates=True)
    for precision in ("fp32", "bf16"):
        m = SceneRF(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8, precision=precision, **kw).to(DEV)
        m.mlp.load_state_dict(mlp)
        m.mlp_gaussian.load_state_dict(mlpg)
        m.debug_aux = precision == "fp32"
        with torch.no_grad():
            out = m.render_rays_batch(K.to(DEV), T.to(DEV), {k: v.to(DEV) for k, v in maps.items()}, sampled_pixels=pix.to(DEV),
                                      ray_batch_size=R, noise=(nu.to(DEV), ng.to(DEV)))
        t = TOL[precision]
        clean = clean_mask(m.last_aux, ref, ocfg, K, R, T, key="larger_chunk") if m.debug_aux else torch.ones(R, dtype=torch.bool)
        _, okd = frac_within(out["depth"].cpu(), ref["depth"].detach(), t["depth"])
        _, okc = frac_within(out["color"].cpu(), ref["color"].detach(), t["color"], True)
        fd, fc = float(okd[clean].float().mean()), float(okc[clean].float().mean())
        print(precision, "rays with oracle-identical indices", int(clean.sum()), "depth frac", fd, "color frac", fc,
              "max rel depth err", float(((out["depth"].cpu() - ref["depth"].detach()).abs() / ref["depth"].detach().abs())[clean].max()),
              "max abs colour err", float((out["color"].cpu() - ref["color"].detach()).abs()[clean].max()))
        assert fd == 1.0 and fc == 1.0


# ------------------------------------------------------------------------------------------------ edge cases
EDGE_CASES = [
    # name, ctor overrides, R, chunk, pose
    ("single_ray", dict(), 1, 1, (2.0, 0.0)),
    ("ragged_chunks", dict(), 37, 16, (1.0, -10.0)),                       # 16 + 16 + 5 rays: last chunk ragged
    ("n96_odd_rows", dict(n_pts_uni=64, n_pts_per_gaussian=8), 7, 7, (5.0, 10.0)),   # M = 672 = 5.25 tiles of 128
    ("gaussians_only", dict(n_pts_uni=0, n_pts_per_gaussian=16), 9, 9, (2.0, 0.0)),  # U = 0 branch, scenerf.py:647-650
    ("two_gaussians", dict(n_gaussians=2, n_pts_per_gaussian=16), 6, 6, (2.0, 10.0)),
    ("behind_camera", dict(), 12, 12, (-30.0, 170.0)),                     # most samples project with z<=0 -> pix (-1,-1)
    ("n512_inference_sampling", dict(n_pts_uni=256, n_pts_per_gaussian=64), 3, 3, (2.0, 0.0)),
]


@pytest.mark.parametrize("name,over,R,chunk,pose", EDGE_CASES, ids=[c[0] for c in EDGE_CASES])
def test_edge_cases_against_oracle_fp32(name, over, R, chunk, pose):
    from scenerf_amd import synth
    kw = dict(sphere_W=376, sphere_H=114)
    kw.update(over)
    ocfg = orc.OracleConfig.kitti(**kw)
    mlp, mlpg = synth.mlp_state(21, 4), synth.mlp_state(22, 2, out_scale=4.0)
    maps = synth.feature_maps(376, 114, 23, smooth=True)
    pix = synth.stride2_pixels((1220, 370), R, 24)
    U, GP = ocfg.n_pts_uni, ocfg.n_gaussians * ocfg.n_pts_per_gaussian
    nu, ng = synth.sampling_noise(R, U, GP, 25)
    K, T = synth.kitti_cam_K(), synth.rel_pose(*pose)
    if U == 0:   # gaussian samples only: the oracle takes empty uniform noise
        nu = torch.zeros(R, 0, 1)
    chunks = [orc.render_chunk(ocfg, mlp, mlpg, K, T, maps, pix[s:s + chunk], nu[s:s + chunk], ng[s:s + chunk], keep_intermediates=True)
              for s in range(0, R, chunk)]
    ref = {k: torch.cat([c[k] for c in chunks], dim=0) for k in OUT_KEYS + ["_idx", "_idx_g"]}
    m = SceneRF(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8, precision="fp32", **kw).to(DEV)
    m.mlp.load_state_dict(mlp)
    m.mlp_gaussian.load_state_dict(mlpg)
    m.debug_aux = True
    x = {k: v.to(DEV).requires_grad_(True) for k, v in maps.items()}
    out = m.render_rays_batch(K.to(DEV), T.to(DEV), x, sampled_pixels=pix.to(DEV), ray_batch_size=chunk,
                              noise=(nu.to(DEV), ng.to(DEV)))
    clean = clean_mask(m.last_aux, ref, ocfg, K, R, T, key="edge_" + name)   # every ray whose indices are the free-running oracle's
    for k in OUT_KEYS:
        got, want = out[k].detach().cpu(), ref[k].detach()
        assert got.shape == want.shape, (k, got.shape, want.shape)
        assert torch.isfinite(got).all(), k
        if k in ("loss_kl", "som_vars", "closest_pts_to_depths", "weights_at_depth"):
            continue   # threshold / argmin outputs: covered by the stage tests with identical inputs
        _, ok = frac_within(got, want, TOL["fp32"]["color"] if k == "color" else 1e-4, k in ABS_KEYS)
        assert bool(ok[clean].all()), "%s: %.3f of the rays within tolerance, max rel error %.2e" % (
            k, float(ok[clean].float().mean()), float(((got - want).abs() / (1 + want.abs())).reshape(R, -1)[clean].max()))
    (out["depth"].sum() + out["color"].sum() + out["loss_kl"].sum()).backward()
    assert all(torch.isfinite(p.grad).all() for p in m.mlp.parameters())
    assert all(torch.isfinite(v.grad).all() for v in x.values())


# ------------------------------------------------------------------------------------------------ caller side
class _StubEncoder(torch.nn.Module):
    """Stands in for the stock EfficientNet-B7 U-Net: fixed maps times a learnable gain (so gradients must arrive)."""

    def __init__(self, maps):
        super().__init__()
        self.gain = torch.nn.Parameter(torch.ones(()))
        self.maps = {k: v.to(DEV) for k, v in maps.items()}

    def forward(self, img, pix=None, pix_sphere=None):
        assert pix.shape == (img.shape[2] * img.shape[3], 2) and pix_sphere.dtype == torch.int64
        return {k: (self.gain * v).unsqueeze(0).expand(img.shape[0], -1, -1, -1) for k, v in self.maps.items()}


def test_training_step_through_the_boundary():
    """Lightning-style call: training_step(batch) -> scalar loss; gradients reach both MLPs and the encoder."""
    from scenerf_amd import synth
    torch.manual_seed(0)
    maps = synth.feature_maps(376, 114, 31, smooth=True)
    m = SceneRF(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8, sphere_W=376, sphere_H=114, n_rays=96,
                net_rgb=_StubEncoder(maps), precision="bf16").to(DEV)
    m.mlp.load_state_dict(synth.mlp_state(32, 4))
    m.mlp_gaussian.load_state_dict(synth.mlp_state(33, 2, out_scale=4.0))
    g = torch.Generator().manual_seed(1)
    img = lambda: torch.rand(3, 370, 1220, generator=g).to(DEV)
    lidar_pix = torch.stack([torch.randint(0, 1220, (50,), generator=g), torch.randint(0, 370, (50,), generator=g)], 1).to(DEV)
    batch = {
        "img_inputs": torch.rand(1, 3, 370, 1220, generator=g).to(DEV),
        "cam_K": synth.kitti_cam_K().unsqueeze(0).to(DEV),
        "T_velo_2_cam": torch.eye(4).unsqueeze(0).to(DEV),
        "T_source2infers": [[synth.rel_pose(2.0, 0.0).to(DEV)]],
        "T_source2targets": [[synth.rel_pose(1.0, 5.0).to(DEV)]],
        "img_sources": [[img()]], "img_targets": [[img()]],
        "loc2d_with_depths": [[lidar_pix]], "lidar_depths": [[(torch.rand(50, generator=g) * 40 + 2).to(DEV)]],
    }
    loss = m.training_step(batch, 0)
    assert loss.ndim == 0 and torch.isfinite(loss)
    loss.backward()
    assert torch.isfinite(m.net_rgb.gain.grad) and float(m.net_rgb.gain.grad.abs()) > 0
    for p in list(m.mlp.parameters()) + list(m.mlp_gaussian.parameters()):
        assert p.grad is not None and torch.isfinite(p.grad).all()
    opt, sched = m.configure_optimizers()
    opt[0].step()
    with torch.no_grad():
        m.validation_step(batch, 0)


def test_training_forward_with_the_fused_source_loss_equals_the_stock_assembly():
    """forward(batch) with the per-source loss assembled by one kernel (scenerf_amd.loss_side.source_loss) against the same forward
    with the loss-side kernel + the stock torch assembly (fused_source_loss = False): same RNG streams, same renderer outputs -> the
    same total, the same logged terms and the same gradient at the encoder."""
    from scenerf_amd import synth
    maps = synth.feature_maps(376, 114, 31, smooth=True)
    res = {}
    for fused in (True, False):
        torch.manual_seed(0)
        m = SceneRF(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8, sphere_W=376, sphere_H=114, n_rays=96,
                    net_rgb=_StubEncoder(maps), precision="fp32").to(DEV)
        m.fused_source_loss = fused
        m.fused_loss_noise = "torch"      # (the reference's randn call in both runs: the device generator then advances identically)
        logged = {}
        m.log = lambda k, v, **kw: logged.__setitem__(k, float(v))
        m.mlp.load_state_dict(synth.mlp_state(32, 4))
        m.mlp_gaussian.load_state_dict(synth.mlp_state(33, 2, out_scale=4.0))
        g = torch.Generator().manual_seed(1)
        img = lambda: torch.rand(3, 370, 1220, generator=g).to(DEV)
        batch = {
            "img_inputs": torch.rand(1, 3, 370, 1220, generator=g).to(DEV),
            "cam_K": synth.kitti_cam_K().unsqueeze(0).to(DEV),
            "T_velo_2_cam": torch.eye(4).unsqueeze(0).to(DEV),
            "T_source2infers": [[synth.rel_pose(2.0, 0.0).to(DEV), synth.rel_pose(3.0, 2.0).to(DEV)]],
            "T_source2targets": [[synth.rel_pose(1.0, 5.0).to(DEV), synth.rel_pose(-1.0, 3.0).to(DEV)]],
            "img_sources": [[img(), img()]], "img_targets": [[img(), img()]],
        }
        torch.manual_seed(5)
        loss = m.training_step(batch, 0)
        loss.backward()
        res[fused] = (float(loss), dict(logged), float(m.net_rgb.gain.grad), m.mlp_gaussian.lin_out.weight.grad.clone())
    (la, ga, ea, wa), (lb, gb, eb, wb) = res[True], res[False]
    assert abs(la - lb) <= 2e-5 * (1 + abs(lb)), (la, lb)
    assert set(ga) == set(gb)
    for k in gb:
        assert abs(ga[k] - gb[k]) <= 2e-5 * (1 + abs(gb[k])), (k, ga[k], gb[k])
    assert abs(ea - eb) <= 1e-3 * abs(eb) + 1e-9, (ea, eb)
    assert float((wa - wb).norm()) <= 1e-3 * float(wb.norm())


def test_optional_output_gradients_match_oracle():
    """weights / alphas / densities / depth_volumes are differentiable outputs like in the reference."""
    from scenerf_amd import synth
    kw = dict(sphere_W=376, sphere_H=114)
    R = 32
    ocfg = orc.OracleConfig.kitti(**kw)
    mlp, mlpg = synth.mlp_state(41, 4), synth.mlp_state(42, 2, out_scale=4.0)
    maps = synth.feature_maps(376, 114, 43, smooth=True)
    pix = synth.stride2_pixels((1220, 370), R, 44)
    nu, ng = synth.sampling_noise(R, 32, 32, 45)
    K, T = synth.kitti_cam_K(), synth.rel_pose(2.0, 5.0)
    gen = torch.Generator().manual_seed(3)
    cw, ca, cd, cz = [torch.randn(R, 64, generator=gen) for _ in range(4)]
    po = {k: v.clone().requires_grad_(True) for k, v in mlp.items()}
    pg = {k: v.clone().requires_grad_(True) for k, v in mlpg.items()}
    ref = orc.render_chunk(ocfg, po, pg, K, T, maps, pix, nu, ng)
    ((ref["weights"] * cw).sum() + (ref["alphas"] * ca).sum() + (ref["densities"] * cd).sum() + (ref["depth_volumes"] * cz).sum()).backward()
    m = SceneRF(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8, precision="fp32", **kw).to(DEV)
    m.mlp.load_state_dict(mlp)
    m.mlp_gaussian.load_state_dict(mlpg)
    out = m.render_rays_batch(K.to(DEV), T.to(DEV), {k: v.to(DEV) for k, v in maps.items()}, sampled_pixels=pix.to(DEV),
                              ray_batch_size=R, noise=(nu.to(DEV), ng.to(DEV)))
    ((out["weights"] * cw.to(DEV)).sum() + (out["alphas"] * ca.to(DEV)).sum() + (out["densities"] * cd.to(DEV)).sum()
     + (out["depth_volumes"] * cz.to(DEV)).sum()).backward()
    for name in ("lin_out.weight", "blocks.1.fc_0.weight", "lin_z.0.weight", "lin_in.weight"):
        got = dict(m.mlp.named_parameters())[name].grad.cpu()
        want = po[name].grad
        rel = float((got - want).norm() / want.norm())
        assert rel < 2e-2, "mlp.%s rel %.3e" % (name, rel)
    for name in ("lin_out.weight", "blocks.2.fc_1.weight"):
        got = dict(m.mlp_gaussian.named_parameters())[name].grad.cpu()
        want = pg[name].grad
        rel = float((got - want).norm() / max(float(want.norm()), 1e-12))
        assert rel < 2e-2, "mlp_gaussian.%s rel %.3e" % (name, rel)


def test_weights_at_depth_and_closest_point_gradients_match_oracle():
    """weights_at_depth = weights[k] and closest_pts_to_depths = |depth - z_k| at k = argmin |depth - z| are differentiable in the
    reference (scenerf.py:729-736: a gather and a min over values with autograd; the index carries none): the same here, through the
    per-ray tail's backward.  Rays whose k differs from the oracle's (a tie in the min) are left out of the loss on both sides."""
    from scenerf_amd import synth
    kw = dict(sphere_W=376, sphere_H=114)
    R = 48
    ocfg = orc.OracleConfig.kitti(**kw)
    mlp, mlpg = synth.mlp_state(61, 4), synth.mlp_state(62, 2, out_scale=4.0)
    maps = synth.feature_maps(376, 114, 63, smooth=True)
    pix = synth.stride2_pixels((1220, 370), R, 64)
    nu, ng = synth.sampling_noise(R, 32, 32, 65)
    K, T = synth.kitti_cam_K(), synth.rel_pose(2.0, 5.0)
    gen = torch.Generator().manual_seed(4)
    cw, cc = torch.randn(R, generator=gen), torch.randn(R, generator=gen)
    po = {k: v.clone().requires_grad_(True) for k, v in mlp.items()}
    pg = {k: v.clone().requires_grad_(True) for k, v in mlpg.items()}
    ref = orc.render_chunk(ocfg, po, pg, K, T, maps, pix, nu, ng, keep_intermediates=True)
    m = SceneRF(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8, precision="fp32", **kw).to(DEV)
    m.mlp.load_state_dict(mlp)
    m.mlp_gaussian.load_state_dict(mlpg)
    m.debug_aux = True
    out = m.render_rays_batch(K.to(DEV), T.to(DEV), {k: v.to(DEV) for k, v in maps.items()}, sampled_pixels=pix.to(DEV),
                              ray_batch_size=R, noise=(nu.to(DEV), ng.to(DEV)))
    assert out["weights_at_depth"].requires_grad and out["closest_pts_to_depths"].requires_grad
    same = (m.last_aux["closest_idx"].cpu().long() == ref["_closest_idx"]).float()
    assert float(same.mean()) > 0.9
    cw, cc = cw * same, cc * same
    ((ref["weights_at_depth"] * cw).sum() + (ref["closest_pts_to_depths"] * cc).sum()).backward()
    ((out["weights_at_depth"] * cw.to(DEV)).sum() + (out["closest_pts_to_depths"] * cc.to(DEV)).sum()).backward()
    for net, ps, names in ((m.mlp, po, ("lin_out.weight", "blocks.1.fc_0.weight", "lin_z.0.weight", "lin_in.weight")),
                           (m.mlp_gaussian, pg, ("lin_out.weight", "blocks.2.fc_1.weight"))):
        for name in names:
            got, want = dict(net.named_parameters())[name].grad.cpu(), ps[name].grad
            assert float(want.norm()) > 0
            rel = float((got - want).norm() / want.norm())
            assert rel < 2e-2, "%s rel %.3e" % (name, rel)


# ------------------------------------------------------------------------------------------------ BASELINE full size
def test_full_size_config2_properties_and_subset_parity():
    """BASELINE.json configs[1] at full size (KITTI 1500x452 sphere, R=1200, N=128): size-independent properties of the
    renderer, plus exact-input parity on a subset -- rays are independent, so the first 40 rays of the full-size GPU run
    must equal the CPU oracle run on just those 40 rays with the same pixels / noise."""
    from scenerf_amd import synth
    R, U, P = 1200, 64, 16
    N = U + 4 * P
    kw = dict(n_pts_uni=U, n_pts_per_gaussian=P)
    mlp, mlpg = synth.mlp_state(51, 4), synth.mlp_state(52, 2, out_scale=4.0)
    maps = synth.feature_maps(1500, 452, 53, smooth=True)
    pix = synth.stride2_pixels((1220, 370), R, 54)
    nu, ng = synth.sampling_noise(R, U, 4 * P, 55)
    K, T = synth.kitti_cam_K(), synth.rel_pose(1.0, 0.0)
    outs = {}
    for precision in ("fp32", "bf16"):
        m = SceneRF(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8, precision=precision, **kw).to(DEV)
        m.mlp.load_state_dict(mlp)
        m.mlp_gaussian.load_state_dict(mlpg)
        m.debug_aux = precision == "fp32"
        with torch.no_grad():
            o = m.render_rays_batch(K.to(DEV), T.to(DEV), {k: v.to(DEV) for k, v in maps.items()}, sampled_pixels=pix.to(DEV),
                                    ray_batch_size=R, noise=(nu.to(DEV), ng.to(DEV)))
        outs[precision] = {k: v.cpu() for k, v in o.items()}
        if precision == "fp32":
            aux32 = {k: m.last_aux[k].cpu() for k in ("sphere_idx", "sphere_idx_g", "dist_sorted", "unit_dir")}
    o = outs["fp32"]
    w, a, z, dep = o["weights"], o["alphas"], o["depth_volumes"], o["depth"]
    assert w.shape == (R, N)
    assert bool((a >= 0).all()) and bool((a <= 1).all()) and bool((w >= 0).all())
    assert bool((w.sum(1) <= 1 + 1e-4).all())                                   # transmittance is a sub-probability
    assert bool((z[:, 1:] >= z[:, :-1] - 1e-6).all())                           # samples sorted along the ray
    assert bool((dep >= 0).all()) and bool((dep <= z.max(dim=1).values + 1e-3).all())   # depth is a convex-ish combination
    T_acc = torch.cumprod(1 - a + 1e-10, dim=1)
    torch.testing.assert_close(w[:, 1:], a[:, 1:] * T_acc[:, :-1], rtol=1e-4, atol=1e-6)   # w_i = a_i * prod_{j<i}(1-a_j)
    torch.testing.assert_close(dep, (w * z).sum(1), rtol=1e-4, atol=1e-4)
    assert bool((o["gaussian_means"] >= 1.5).all()) and bool((o["gaussian_stds"] >= 1.5).all())   # relu(.) + 1.5 floors
    assert bool((o["densities"] >= 0).all()) and bool((o["color"] >= 0).all()) and bool((o["color"] <= 1 + 1e-4).all())
    # bf16 operands vs fp32 operands at full size
    rel = ((outs["bf16"]["depth"] - dep).abs() / dep.abs().clamp(min=1e-3))
    assert float(rel.median()) < 5e-3 and float(rel.quantile(0.99)) < 5e-2, (float(rel.median()), float(rel.max()))
    assert float((outs["bf16"]["color"] - o["color"]).abs().max()) < 5e-2
    # subset parity against the oracle (CPU) on identical inputs
    S = 40
    ocfg = orc.OracleConfig.kitti(**kw)
    ref = orc.render_chunk(ocfg, mlp, mlpg, K, T, maps, pix[:S], nu[:S], ng[:S], keep_intermediates=True)
    clean = clean_mask(aux32, ref, ocfg, K, S, T, key="full_config2_subset")
    for k in ("depth", "color", "weights", "alphas", "densities", "gaussian_means", "gaussian_stds", "depth_volumes"):
        _, ok = frac_within(o[k][:S], ref[k].detach(), TOL["fp32"]["color"] if k == "color" else 1e-4, k in ABS_KEYS)
        assert bool(ok[clean].all()), "%s: %.3f of the subset rays within tolerance" % (k, float(ok[clean].float().mean()))


@pytest.mark.gpu
def test_inference_chunk_replays_as_a_graph():
    """Config C5's mechanics (static chunk, no_grad, hipGraph): with the sampling noise supplied, one chunk of
    render_rays_batch is launch-only (no host sync, no rocSOLVER after the first call, allocations inside the graph pool), so it
    captures into a graph whose replay reproduces the eager outputs bit for bit -- on the lean fused-inference path (only H3 is
    written; the other activations and the sign bits are NULL)."""
    from scenerf_amd import synth
    kw = dict(sphere_W=376, sphere_H=114, n_pts_uni=32, n_pts_per_gaussian=8)
    R = 512   # x 64 samples = 32768 rows >= 4096: fused kernel
    m = SceneRF(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8, precision="bf16", **kw).to(DEV).eval()
    m.mlp.load_state_dict(synth.mlp_state(21, 4))
    m.mlp_gaussian.load_state_dict(synth.mlp_state(22, 2, out_scale=4.0))
    maps = {k: v.to(DEV) for k, v in synth.feature_maps(376, 114, 23, smooth=True).items()}
    K, T = synth.kitti_cam_K().to(DEV), synth.rel_pose(2.0, 10.0).to(DEV)
    pix = synth.stride2_pixels((1220, 370), R, 24).to(DEV)
    nu, ng = synth.sampling_noise(R, 32, 32, 25)
    nu, ng = nu.to(DEV), ng.to(DEV)
    with torch.no_grad():
        ref = m.render_rays_batch(K, T, maps, sampled_pixels=pix, ray_batch_size=R, noise=(nu, ng))
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            m.render_rays_batch(K, T, maps, sampled_pixels=pix, ray_batch_size=R, noise=(nu, ng))
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = m.render_rays_batch(K, T, maps, sampled_pixels=pix, ray_batch_size=R, noise=(nu, ng))
        for _ in range(2):
            g.replay()
        torch.cuda.synchronize()
    for k in OUT_KEYS:
        assert torch.equal(ref[k], out[k]), k
    # and the lean path agrees with the training-mode forward (all activations saved) on the same inputs
    maps_g = {k: v.clone().requires_grad_(True) for k, v in maps.items()}
    full = m.render_rays_batch(K, T, maps_g, sampled_pixels=pix, ray_batch_size=R, noise=(nu, ng))
    for k in ("depth", "color", "loss_kl"):
        assert torch.equal(ref[k], full[k].detach()), k


@pytest.mark.gpu
def test_full_size_config3_bundlefusion_fused_vs_layers_and_subset_parity():
    """BASELINE.json configs[3] at full size (BundleFusion 640x480, sphere 960x720 from the CLI, R=1080, N=96, D=12, std 0.1,
    floors +0.5), a different scale-activity pattern and K-segment mix than KITTI for the fused kernels.  (1) fp32 forward: the first 24 rays equal the CPU oracle run on just those rays (rays are independent).
    (2) bf16 training step on the fused kernels vs the same step on the per-layer kernels (render_cfg.fused_min_rows = -1): outputs within bf16
    rounding, every parameter / feature-map gradient within 2e-2 relative L2 (the dgrad chain itself is bit-identical; the
    forward's residual stream is rounded at the same places but accumulated in a different order)."""
    from scenerf_amd import synth
    R, U, P = 1080, 64, 8
    kw = dict(som_sigma=0.02, std=0.1, add_fov_hor=14, add_fov_ver=11, sphere_W=960, sphere_H=720, n_pts_uni=U, n_pts_per_gaussian=P,
              max_sample_depth=12)
    mlp, mlpg = synth.mlp_state(61, 4), synth.mlp_state(62, 2, out_scale=0.5)
    maps = synth.feature_maps(960, 720, 63, smooth=True)
    pix = synth.stride2_pixels((640, 480), R, 64)
    nu, ng = synth.sampling_noise(R, U, 4 * P, 65)
    K, T = synth.bundlefusion_cam_K(), synth.rel_pose(0.3, 8.0)

    auxs = {}

    def run(precision, grad, fused=True):
        m = SceneRFBundleFusion(precision=precision, **kw).to(DEV)
        m.debug_aux = precision == "fp32"
        if not fused:
            m.render_cfg.fused_min_rows = -1   # per-layer GEMM path (explicit call state: scenerf_cfg.fused_min_rows)
        m.mlp.load_state_dict(mlp)
        m.mlp_gaussian.load_state_dict(mlpg)
        x = {k: v.to(DEV).requires_grad_(grad) for k, v in maps.items()}
        with torch.set_grad_enabled(grad):
            o = m.render_rays_batch(K.to(DEV), T.to(DEV), x, sampled_pixels=pix.to(DEV), ray_batch_size=R, noise=(nu.to(DEV), ng.to(DEV)))
        grads = None
        if m.debug_aux:
            auxs[precision] = {k: m.last_aux[k].cpu() for k in ("sphere_idx", "sphere_idx_g", "dist_sorted", "unit_dir")}
        if grad:
            (o["depth"].mean() + o["color"].mean() + o["loss_kl"].mean() + o["gaussian_means"].mean()).backward()
            grads = {"mlp." + n: p.grad.cpu() for n, p in m.mlp.named_parameters()}
            grads.update({"mlpg." + n: p.grad.cpu() for n, p in m.mlp_gaussian.named_parameters()})
            grads.update({"map." + k: (v.grad.cpu() if v.grad is not None else torch.zeros_like(v).cpu()) for k, v in x.items()})
        return {k: v.detach().cpu() for k, v in o.items()}, grads

    # (1) fp32 subset parity
    o32, _ = run("fp32", False)
    S = 24
    ocfg = orc.OracleConfig.bundlefusion(**{k: v for k, v in kw.items()})
    ref = orc.render_chunk(ocfg, mlp, mlpg, K, T, maps, pix[:S], nu[:S], ng[:S], keep_intermediates=True)
    clean = clean_mask(auxs["fp32"], ref, ocfg, K, S, T, key="full_config3_subset")
    for k in ("depth", "color", "weights", "alphas", "gaussian_means", "gaussian_stds", "depth_volumes"):
        _, ok = frac_within(o32[k][:S], ref[k].detach(), TOL["fp32"]["color"] if k == "color" else 1e-4, k in ABS_KEYS)
        assert bool(ok[clean].all()), "%s: %.3f of the subset rays within tolerance" % (k, float(ok[clean].float().mean()))
    assert bool((o32["gaussian_means"] >= 0.5).all()) and bool((o32["gaussian_stds"] >= 0.5).all())   # scenerf_bf.py:606-608
    # (2) fused vs per-layer kernels, bf16, forward + backward
    ol, gl = run("bf16", True, fused=False)
    of, gf = run("bf16", True)
    rel = (of["depth"] - ol["depth"]).abs() / ol["depth"].abs().clamp(min=1e-3)
    assert float(rel.median()) < 2e-3 and float(rel.quantile(0.99)) < 3e-2, (float(rel.median()), float(rel.max()))
    assert float((of["color"] - ol["color"]).abs().max()) < 3e-2
    touched = 0
    for n in gl:
        a, b = gl[n].double(), gf[n].double()
        if float(a.norm()) == 0.0:
            assert float(b.norm()) == 0.0, n          # (scales no ray reaches get exactly zero gradient on both paths)
            continue
        touched += n.startswith("map.")
        r = float((a - b).norm() / a.norm())
        assert r <= 2e-2, "%s: fused vs layers relative L2 %.3e" % (n, r)
    assert touched >= 2, "at least the two finest pyramid scales receive gradient (quirk Q1 keeps the coarse ones out of range)"


@pytest.mark.gpu
def test_chunked_call_equals_single_chunk_on_the_fused_path():
    """render_rays_batch in two chunks of 128 rays (8,192 rows each: fused kernels) against one chunk of 256 rays with the same pixels
    and noise: rays are independent, so the outputs must agree to bf16-kernel reproducibility, and the gradients -- accumulated
    across the chunks in the same packed sinks / map accumulators -- to atomic-ordering noise."""
    from scenerf_amd import synth
    kw = dict(sphere_W=376, sphere_H=114, n_pts_uni=32, n_pts_per_gaussian=8)
    R = 256
    mlp, mlpg = synth.mlp_state(81, 4), synth.mlp_state(82, 2, out_scale=4.0)
    maps = synth.feature_maps(376, 114, 83, smooth=True)
    pix = synth.stride2_pixels((1220, 370), R, 84).to(DEV)
    nu, ng = synth.sampling_noise(R, 32, 32, 85)
    nu, ng = nu.to(DEV), ng.to(DEV)
    K, T = synth.kitti_cam_K().to(DEV), synth.rel_pose(2.0, 10.0).to(DEV)
    res = {}
    for chunk in (R, R // 2):
        m = SceneRF(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8, precision="bf16", **kw).to(DEV)
        m.mlp.load_state_dict(mlp)
        m.mlp_gaussian.load_state_dict(mlpg)
        x = {k: v.to(DEV).requires_grad_(True) for k, v in maps.items()}
        out = m.render_rays_batch(K, T, x, sampled_pixels=pix, ray_batch_size=chunk, noise=(nu, ng))
        (out["depth"].mean() + out["color"].mean() + out["loss_kl"].mean() + out["gaussian_means"].mean()).backward()
        g = {"mlp." + n: p.grad for n, p in m.mlp.named_parameters()}
        g.update({"mlpg." + n: p.grad for n, p in m.mlp_gaussian.named_parameters()})
        g.update({"map." + k: v.grad for k, v in x.items()})
        res[chunk] = ({k: v.detach() for k, v in out.items()}, g)
    (o1, g1), (o2, g2) = res[R], res[R // 2]
    for k in ("depth", "color", "loss_kl", "gaussian_means", "weights"):
        torch.testing.assert_close(o2[k], o1[k], rtol=1e-3, atol=1e-3, msg=lambda s_, k=k: "%s: %s" % (k, s_))
    for n in g1:
        a, b = g1[n].double(), g2[n].double()
        if float(a.norm()) == 0.0:
            assert float(b.norm()) == 0.0, n
            continue
        r = float((a - b).norm() / a.norm())
        assert r <= 2e-3, "%s: chunked vs single relative L2 %.3e" % (n, r)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_hwc_entry_equals_the_chw_entry(precision):
    """x_rgb handed over as fp32 (H, W, C) tensors (renderer.HWC: read in place, gradient returned in (H, W, C)) against the reference's
    (C, H, W) entry on the same values: fp32 mode reads the same numbers through the same arithmetic -- outputs bit-identical, gradients
    to atomic-ordering noise; bf16 mode blends unrounded taps (the (C, H, W) entry rounds the maps to bf16 first): bf16-level agreement.
    Levels can be mixed (here the 1/16 level stays (C, H, W))."""
    from scenerf_amd import synth
    from scenerf_amd.renderer import HWC
    kw = dict(sphere_W=376, sphere_H=114, n_pts_uni=32, n_pts_per_gaussian=8)
    R = 160
    mlp, mlpg = synth.mlp_state(91, 4), synth.mlp_state(92, 2, out_scale=4.0)
    maps = synth.feature_maps(376, 114, 93, smooth=True)
    pix = synth.stride2_pixels((1220, 370), R, 94).to(DEV)
    nu, ng = synth.sampling_noise(R, 32, 32, 95)
    nu, ng = nu.to(DEV), ng.to(DEV)
    K, T = synth.kitti_cam_K().to(DEV), synth.rel_pose(2.0, 10.0).to(DEV)
    res = {}
    for entry in ("chw", "hwc", "cl"):
        m = SceneRF(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8, precision=precision, **kw).to(DEV)
        m.mlp.load_state_dict(mlp)
        m.mlp_gaussian.load_state_dict(mlpg)
        leaves = {}
        x = {}
        for k, v in maps.items():
            if entry == "hwc" and k != "1_16":
                leaves[k] = v.permute(1, 2, 0).contiguous().to(DEV).requires_grad_(True)
                x[k] = HWC(leaves[k])
            elif entry == "cl" and k != "1_16":   # no wrapper: a (C,H,W) tensor with channels-last strides (a torch.channels_last slice)
                c, h, w = v.shape
                leaves[k] = torch.empty_strided((c, h, w), (1, w * c, c), device=DEV).copy_(v.to(DEV)).requires_grad_(True)
                x[k] = leaves[k]
            else:
                leaves[k] = v.to(DEV).requires_grad_(True)
                x[k] = leaves[k]
        out = m.render_rays_batch(K, T, x, sampled_pixels=pix, ray_batch_size=R, noise=(nu, ng))
        (out["depth"].mean() + out["color"].mean() + out["loss_kl"].mean() + out["gaussian_means"].mean()).backward()
        g = {"mlp." + n: p.grad for n, p in m.mlp.named_parameters()}
        g.update({"mlpg." + n: p.grad for n, p in m.mlp_gaussian.named_parameters()})
        for k, v in leaves.items():
            assert v.grad is not None and v.grad.shape == v.shape
            g["map." + k] = v.grad.permute(2, 0, 1) if (entry == "hwc" and k != "1_16") else v.grad
        res[entry] = ({k: v.detach() for k, v in out.items()}, g)
    (oh, gh), (oc, gc) = res["hwc"], res["cl"]
    for k in oh:       # the wrapper and the stride detection are the same path
        assert torch.equal(oh[k], oc[k]), k
    for n in gh:
        assert float((gh[n].double() - gc[n].double()).norm()) <= 2e-5 * float(gh[n].double().norm()) + 1e-30, n
    (o1, g1), (o2, g2) = res["chw"], res["hwc"]
    for k in ("depth", "color", "loss_kl", "gaussian_means", "weights"):
        if precision == "fp32":
            assert torch.equal(o2[k], o1[k]), k
        else:
            torch.testing.assert_close(o2[k], o1[k], rtol=2e-2, atol=2e-2, msg=lambda s_, k=k: "%s: %s" % (k, s_))
    tol = 2e-5 if precision == "fp32" else 6e-2
    for n in g1:
        a, b = g1[n].double(), g2[n].double()
        if float(a.norm()) == 0.0:
            assert float(b.norm()) == 0.0, n
            continue
        r = float((a - b).norm() / a.norm())
        assert r <= tol, "%s: HWC vs CHW entry relative L2 %.3e" % (n, r)
    with pytest.raises(RuntimeError, match="HWC"):   # a (C, H, W) tensor inside the wrapper is refused, not reinterpreted
        m.render_rays_batch(K, T, {k: HWC(v.to(DEV)) for k, v in maps.items()}, sampled_pixels=pix, ray_batch_size=R, noise=(nu, ng))


class _BasicBlock(torch.nn.Module):
    """Shape of the reference's residual block (unet2d_sphere.py:9-34): conv-BN-LeakyReLU, conv-BN, + residual, LeakyReLU."""

    def __init__(self, c, d):
        super().__init__()
        nn = torch.nn
        self.b1 = nn.Sequential(nn.Conv2d(c, c, 3, padding=d, dilation=d), nn.BatchNorm2d(c), nn.LeakyReLU())
        self.b2 = nn.Sequential(nn.Conv2d(c, c, 3, padding=d, dilation=d), nn.BatchNorm2d(c))
        self.act = nn.LeakyReLU()

    def forward(self, x):
        return self.act(self.b2(self.b1(x)) + x)


def _upsample_bn_like(c_in, c_out):
    """The tail of the decoder that emits x_rgb (UpSampleBN._net, unet2d_sphere.py:37-56): 3x3 conv + three dilated residual blocks."""
    return torch.nn.Sequential(torch.nn.Conv2d(c_in, c_out, 3, padding=1), _BasicBlock(c_out, 1), _BasicBlock(c_out, 2), _BasicBlock(c_out, 3))


@pytest.mark.gpu
def test_channels_last_decoder_stack_is_read_in_place():
    """What bench.py's default entry assumes: a stock conv stack run in torch.channels_last (MIOpen convolutions, BatchNorm, LeakyReLU,
    residual adds -- the layers of the reference's UpSampleBN) emits channels-last memory, a slice ``out[k][i]`` of its (B,C,H,W) output
    is a (C,H,W) tensor whose memory is (H,W,C), and render_rays_batch reads it in place: layout code 2 for the levels it converts
    otherwise, no layout-conversion kernel in the step (in-library launch table), outputs bit-identical to the contiguous entry in fp32
    and the gradients that reach the decoder's parameters equal to atomic-ordering noise."""
    from scenerf_amd import _capi, synth
    from scenerf_amd.config import FEAT_CHANNELS
    from scenerf_amd.renderer import RenderSession
    kw = dict(sphere_W=376, sphere_H=114, n_pts_uni=32, n_pts_per_gaussian=8)
    R, B = 160, 2
    mlp, mlpg = synth.mlp_state(91, 4), synth.mlp_state(92, 2, out_scale=4.0)
    pix = synth.stride2_pixels((1220, 370), R, 94).to(DEV)
    nu, ng = synth.sampling_noise(R, 32, 32, 95)
    nu, ng = nu.to(DEV), ng.to(DEV)
    K, T = synth.kitti_cam_K().to(DEV), synth.rel_pose(2.0, 10.0).to(DEV)
    m = SceneRF(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8, precision="fp32", **kw).to(DEV)
    m.mlp.load_state_dict(mlp)
    m.mlp_gaussian.load_state_dict(mlpg)
    shapes = m.render_cfg.map_shapes()
    torch.manual_seed(5)
    nets = {s: _upsample_bn_like(8, c).to(DEV) for s, c in zip((1, 2, 4, 8, 16), FEAT_CHANNELS)}
    ins = {s: torch.randn(B, 8, h, w, device=DEV) for s, (c, h, w) in zip((1, 2, 4, 8, 16), shapes)}
    lib = _capi.load()
    res = {}
    for entry in ("channels_last", "contiguous"):
        for net in nets.values():
            net.zero_grad(set_to_none=True)
            net.to(memory_format=torch.channels_last if entry == "channels_last" else torch.contiguous_format)
        outs = {}
        for s, net in nets.items():
            x = ins[s].to(memory_format=torch.channels_last) if entry == "channels_last" else ins[s].contiguous()
            outs[s] = net(x)
        x_rgb = {"1_%d" % s: (o[1] if entry == "channels_last" else o[1].contiguous()) for s, o in outs.items()}
        if entry == "channels_last":
            for s, o in outs.items():
                assert o.is_contiguous(memory_format=torch.channels_last), "level 1/%d: the conv stack did not keep channels_last" % s
                c, h, w = x_rgb["1_%d" % s].shape
                assert x_rgb["1_%d" % s].stride() == (1, w * c, c)
            assert RenderSession.classify_maps(x_rgb)[0] == (0, 1, 2, 3, 4)
        torch.cuda.synchronize()
        lib.scenerf_hip_profile_enable(1)
        out = m.render_rays_batch(K, T, x_rgb, sampled_pixels=pix, ray_batch_size=R, noise=(nu, ng))
        (out["depth"].mean() + out["color"].mean() + out["loss_kl"].mean() + out["gaussian_means"].mean()).backward()
        torch.cuda.synchronize()
        names = {k["name"] for k in _capi.profile_collect()}
        lib.scenerf_hip_profile_enable(0)
        if entry == "channels_last":
            assert not (names & {"maps_chw_to_hwc", "grads_hwc_to_chw"}), names
        else:
            assert {"maps_chw_to_hwc", "grads_hwc_to_chw"} <= names
        grads = {"%d.%s" % (s, n): p.grad.detach().clone().contiguous() for s, net in nets.items() for n, p in net.named_parameters()}
        res[entry] = ({k: v.detach().clone() for k, v in out.items()}, grads)
    (o1, g1), (o2, g2) = res["channels_last"], res["contiguous"]
    # the conv stack itself may pick different MIOpen kernels per layout: compare what the renderer was GIVEN first
    for k in ("depth", "color", "loss_kl", "gaussian_means", "weights"):
        torch.testing.assert_close(o1[k], o2[k], rtol=1e-4, atol=1e-5, msg=lambda s_, k=k: "%s: %s" % (k, s_))
    # (a conv bias in front of a BatchNorm has a mathematically zero gradient -- what arrives is rounding noise on both sides -- so the
    # error of a tensor is taken relative to the largest gradient of its level's stack, not to its own norm)
    scale = {}
    for n, b in g2.items():
        lvl = n.split(".")[0]
        scale[lvl] = max(scale.get(lvl, 0.0), float(b.double().norm()))
    worst = 0.0
    for n in g1:
        a, b = g1[n].double(), g2[n].double()
        lvl = n.split(".")[0]
        if scale[lvl] == 0.0:   # levels no sample reaches (quirk Q1)
            assert float(a.norm()) == 0.0, n
            continue
        r = float((a - b).norm() / max(float(b.norm()), 1e-2 * scale[lvl]))
        worst = max(worst, r)
        # (MIOpen picks different convolution algorithms per layout: the two conv stacks themselves differ at the 1e-3 level in their
        # parameter gradients -- measured worst 2.4e-3 -- before the renderer's entry plays any role)
        assert r <= 1e-2, "%s: decoder-parameter gradient through the channels-last entry vs the contiguous one: rel L2 %.3e" % (n, r)
    print("channels-last decoder stack: worst decoder-parameter gradient difference vs the contiguous entry %.2e" % worst)
    assert scale["1"] > 0 and scale["2"] > 0


# ------------------------------------------------------------------------------------------------ full-frame inference (C5)
@pytest.mark.gpu
def test_render_image_static_chunks_and_graph_replay():
    """SceneRF.render_image (scenerf_amd/inference.py): the tail chunk is padded to the static chunk size and one captured hipGraph
    is replayed per chunk.  (1) graph replay == the same static chunks launched eagerly, bit for bit; (2) a no_grad multi-chunk
    render_rays_batch call (what render_colors.py / generate_novel_depths.py issue) takes that route by itself; (3) against the
    reference-style ragged chunk loop: identical for the full chunks, within bf16 rounding for the rays of the tail chunk (padded
    to 8,192 rows it runs on the fused kernels, ragged at 2,816 rows on the per-layer ones)."""
    from scenerf_amd import synth
    kw = dict(sphere_W=376, sphere_H=114, n_pts_uni=32, n_pts_per_gaussian=8)
    R, CH = 300, 128
    m = SceneRF(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8, precision="bf16", **kw).to(DEV).eval()
    m.mlp.load_state_dict(synth.mlp_state(91, 4))
    m.mlp_gaussian.load_state_dict(synth.mlp_state(92, 2, out_scale=4.0))
    maps = {k: v.to(DEV) for k, v in synth.feature_maps(376, 114, 93, smooth=True).items()}
    K, T = synth.kitti_cam_K().to(DEV), synth.rel_pose(2.0, 10.0).to(DEV)
    pix = synth.stride2_pixels((1220, 370), R, 94).to(DEV)
    nu, ng = synth.sampling_noise(R, 32, 32, 95)
    noise = (nu.to(DEV), ng.to(DEV))
    with torch.no_grad():
        a = m.render_image(K, T, maps, sampled_pixels=pix, ray_batch_size=CH, noise=noise, use_graph=True)
        eng = m._image_renderer[1]
        assert eng.graph is not None and eng.replays == 3
        a2 = m.render_image(K, T, maps, sampled_pixels=pix, ray_batch_size=CH, noise=noise, use_graph=True)   # same frame: same engine, same graph
        assert m._image_renderer[1] is eng and eng.replays == 6
        b = m.render_image(K, T, maps, sampled_pixels=pix, ray_batch_size=CH, noise=noise, use_graph=False)
        c = m.render_rays_batch(K, T, maps, sampled_pixels=pix, ray_batch_size=CH, noise=noise)                # routed through render_image
        m.static_inference = False
        d = m.render_rays_batch(K, T, maps, sampled_pixels=pix, ray_batch_size=CH, noise=noise)                # the reference's ragged loop
    assert set(a) == set(OUT_KEYS)
    for k in OUT_KEYS:
        assert a[k].shape[0] == R
        assert torch.equal(a[k], a2[k]) and torch.equal(a[k], b[k]) and torch.equal(a[k], c[k]), k
        assert torch.equal(a[k][:2 * CH], d[k][:2 * CH]), k
    rel = (a["depth"][2 * CH:] - d["depth"][2 * CH:]).abs() / d["depth"][2 * CH:].abs()
    assert float(rel.max()) < 3e-2, float(rel.max())
    # device RNG (no noise given): runs, finite, and differs from call to call (inference is stochastic in the reference too)
    with torch.no_grad():
        m.static_inference = True
        e1 = m.render_image(K, T, maps, sampled_pixels=pix, ray_batch_size=CH, keys=("depth", "color"))
        e2 = m.render_image(K, T, maps, sampled_pixels=pix, ray_batch_size=CH, keys=("depth", "color"))
    assert set(e1) == {"depth", "color"} and bool(torch.isfinite(e1["depth"]).all()) and not torch.equal(e1["depth"], e2["depth"])


@pytest.mark.gpu
def test_render_image_follows_in_place_writes_no_version_counter_sees():
    """The cached full-frame engine (packed weights, converted maps, a captured graph) must not serve stale inputs: ``p.data.copy_``
    and in-place map writes through ``.data`` bump no version counter, moving a parameter (``p.data = ...``) changes its address."""
    from scenerf_amd import synth
    kw = dict(sphere_W=376, sphere_H=114, n_pts_uni=32, n_pts_per_gaussian=8)
    R, CH = 300, 128
    m = SceneRF(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8, precision="bf16", **kw).to(DEV).eval()
    m.mlp.load_state_dict(synth.mlp_state(91, 4))
    m.mlp_gaussian.load_state_dict(synth.mlp_state(92, 2, out_scale=4.0))
    maps = {k: v.to(DEV) for k, v in synth.feature_maps(376, 114, 93, smooth=True).items()}
    K, T = synth.kitti_cam_K().to(DEV), synth.rel_pose(2.0, 10.0).to(DEV)
    pix = synth.stride2_pixels((1220, 370), R, 94).to(DEV)
    nu, ng = synth.sampling_noise(R, 32, 32, 95)
    noise = (nu.to(DEV), ng.to(DEV))

    def render():
        with torch.no_grad():
            return m.render_image(K, T, maps, sampled_pixels=pix, ray_batch_size=CH, noise=noise, use_graph=True, keys=("depth", "color"))

    def fresh():
        m.release_inference_engine()
        return render()

    a = render()
    eng = m._image_renderer[1]
    new_w = synth.mlp_state(191, 4)
    for n, p_ in m.mlp.named_parameters():
        v0 = p_._version
        p_.data.copy_(new_w[n].to(DEV))            # e.g. an EMA swap: no version bump, same address
        assert p_._version == v0
    b = render()
    assert m._image_renderer[1] is eng              # same engine, same graph ...
    assert not torch.equal(a["depth"], b["depth"])  # ... new weights
    b_ref = fresh()
    assert torch.equal(b["depth"], b_ref["depth"]) and torch.equal(b["color"], b_ref["color"])
    eng = m._image_renderer[1]
    maps["1_1"].data.mul_(0.5)                      # a map rewritten in place through .data
    c = render()
    assert m._image_renderer[1] is eng
    c_ref = fresh()
    assert not torch.equal(b["color"], c["color"]) and torch.equal(c["color"], c_ref["color"]) and torch.equal(c["depth"], c_ref["depth"])
    eng = m._image_renderer[1]
    m.mlp.lin_out.weight.data = m.mlp.lin_out.weight.data.clone() * 2.0   # a parameter that MOVED: the engine is rebuilt
    d = render()
    assert m._image_renderer[1] is not eng
    assert not torch.equal(c["color"], d["color"])


@pytest.mark.gpu
def test_no_grad_multi_chunk_call_keeps_the_reference_rng_stream():
    """A seeded evaluation run must see the same samples whether or not render_rays_batch routes through the static-chunk engine:
    with device_rng=False (default) the gaussian noise comes from the CPU generator chunk by chunk, shapes and order as in
    scenerf.py:437-455 / utils.py:208-211; the uniform noise from the device generator (utils.py:84)."""
    from scenerf_amd import synth
    kw = dict(sphere_W=376, sphere_H=114, n_pts_uni=32, n_pts_per_gaussian=8)
    R, CH = 300, 128
    m = SceneRF(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8, precision="fp32", **kw).to(DEV).eval()
    m.mlp.load_state_dict(synth.mlp_state(91, 4))
    m.mlp_gaussian.load_state_dict(synth.mlp_state(92, 2, out_scale=4.0))
    maps = {k: v.to(DEV) for k, v in synth.feature_maps(376, 114, 93, smooth=True).items()}
    K, T = synth.kitti_cam_K().to(DEV), synth.rel_pose(2.0, 10.0).to(DEV)
    pix = synth.stride2_pixels((1220, 370), R, 94).to(DEV)
    outs = []
    for static in (True, False, True):
        m.static_inference = static
        torch.manual_seed(1234)   # CPU and device generators
        with torch.no_grad():
            outs.append(m.render_rays_batch(K, T, maps, sampled_pixels=pix, ray_batch_size=CH))
    for k in OUT_KEYS:
        assert torch.equal(outs[0][k], outs[1][k]), k
        assert torch.equal(outs[0][k], outs[2][k]), k


@pytest.mark.gpu
def test_render_image_n512_bf16_against_position_matched_oracle():
    """BASELINE.json configs[4] sampling (N = 512: U=256, G=4, P=64), bf16, one padded chunk of 32 rays = 16,384 rows on the fused
    lean-inference path, replayed from the captured graph, white-noise maps at the full KITTI sphere: depth and colour of every ray
    against the CPU oracle evaluated at the GPU's own gaussian-head offsets (render_chunk(head_offsets=...): identical sample
    positions, so the comparison is bf16 arithmetic, not samples crossing texel boundaries)."""
    from scenerf_amd import synth
    U, P, R = 256, 64, 20
    kw = dict(n_pts_uni=U, n_pts_per_gaussian=P)
    mlp, mlpg = synth.mlp_state(101, 4), synth.mlp_state(102, 2, out_scale=4.0)
    maps = synth.feature_maps(1500, 452, 103, smooth=False)
    pix = synth.stride2_pixels((1220, 370), R, 104)
    nu, ng = synth.sampling_noise(R, U, 4 * P, 105)
    K, T = synth.kitti_cam_K(), synth.rel_pose(2.0, 5.0)
    m = SceneRF(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8, precision="bf16", **kw).to(DEV).eval()
    m.mlp.load_state_dict(mlp)
    m.mlp_gaussian.load_state_dict(mlpg)
    m.debug_aux = True
    with torch.no_grad():
        out = m.render_image(K.to(DEV), T.to(DEV), {k: v.to(DEV) for k, v in maps.items()}, sampled_pixels=pix.to(DEV), ray_batch_size=32,
                             noise=(nu.to(DEV), ng.to(DEV)), use_graph=True)
    eng = m._image_renderer[1]
    assert eng.graph is not None and m.render_cfg.uses_fused(32 * 512)
    off = eng.session.last_aux["offsets"].float().cpu().reshape(32, 4, 2)[:R]
    ref = orc.render_chunk(orc.OracleConfig.kitti(**kw), mlp, mlpg, K, T, maps, pix, nu, ng, head_offsets=off)
    rel = (out["depth"].cpu() - ref["depth"].detach()).abs() / ref["depth"].detach().abs()
    cerr = (out["color"].cpu() - ref["color"].detach()).abs()
    print("N=512 bf16 render_image vs matched oracle: depth rel max %.2e median %.2e, colour abs max %.2e" % (
        float(rel.max()), float(rel.median()), float(cerr.max())))
    assert torch.equal(out["gaussian_means"].cpu(), ref["gaussian_means"].detach())
    assert float(rel.max()) < 3e-2 and float(cerr.max()) < 3e-2


@pytest.mark.gpu
def test_render_image_n512_at_the_benched_chunk_against_the_oracle():
    """BASELINE.json configs[4] AS BENCHED (bench.py's infer_c5 leg): N = 512 (U=256, G=4, P=64), bf16, ONE chunk of 4,096 rays =
    2,097,152 rows = 16,384 row blocks of the 128-row forward, `keys=("depth", "color")` -- the compositing-only instantiation of the
    per-ray tail (ray_tail_fwd_kernel<8, 4, false>, selective outputs), replayed from the captured graph -- against the CPU oracle under
    no_grad, evaluated at the GPU's own gaussian-head offsets (identical sample positions: the comparison is the arithmetic of the
    radiance MLP and the compositing, at the chunk shape the bench line is quoted on).  The oracle walks the chunk 128 rays at a time
    (its gathered features are 10 KB per sample in fp32) over every ORACLE_STRIDE-th ray: 1,024 rays spread over all of the chunk."""
    from scenerf_amd import synth
    U, P, R = 256, 64, 4096
    ORACLE_STRIDE = 4
    kw = dict(n_pts_uni=U, n_pts_per_gaussian=P)
    mlp, mlpg = synth.mlp_state(111, 4), synth.mlp_state(112, 2, out_scale=4.0)
    maps = synth.feature_maps(1500, 452, 113, smooth=False)
    pix = synth.stride2_pixels((1220, 370), R, 114)
    nu, ng = synth.sampling_noise(R, U, 4 * P, 115)
    K, T = synth.kitti_cam_K(), synth.rel_pose(2.0, 5.0)
    m = SceneRF(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8, precision="bf16", **kw).to(DEV).eval()
    m.mlp.load_state_dict(mlp)
    m.mlp_gaussian.load_state_dict(mlpg)
    m.debug_aux = True
    with torch.no_grad():
        out = m.render_image(K.to(DEV), T.to(DEV), {k: v.to(DEV) for k, v in maps.items()}, sampled_pixels=pix.to(DEV), ray_batch_size=R,
                             keys=("depth", "color"), noise=(nu.to(DEV), ng.to(DEV)), use_graph=True)
    assert set(out) == {"depth", "color"}
    eng = m._image_renderer[1]
    assert eng.graph is not None and m.render_cfg.uses_fused(R * (U + 4 * P))
    off = eng.session.last_aux["offsets"].float().cpu().reshape(R, 4, 2)
    sel = torch.arange(0, R, ORACLE_STRIDE)
    ocfg = orc.OracleConfig.kitti(index_rule="pinned", **kw)
    dref, cref = [], []
    import time
    t0 = time.time()
    with torch.no_grad():
        for s in range(0, len(sel), 128):
            ii = sel[s:s + 128]
            o = orc.render_chunk(ocfg, mlp, mlpg, K, T, maps, pix[ii], nu[ii], ng[ii], head_offsets=off[ii])
            dref.append(o["depth"]); cref.append(o["color"])
    dref, cref = torch.cat(dref), torch.cat(cref)
    rel = (out["depth"].cpu()[sel] - dref).abs() / dref.abs()
    cerr = (out["color"].cpu()[sel] - cref).abs()
    print("N=512, one 4,096-ray chunk, depth + colour only, bf16 vs the oracle at matched positions (%d rays, oracle %.0f s): depth rel max %.2e "
          "median %.2e p99 %.2e, colour abs max %.2e p99 %.2e" % (len(sel), time.time() - t0, float(rel.max()), float(rel.median()),
                                                                 float(rel.quantile(0.99)), float(cerr.max()), float(cerr.quantile(0.99))))
    # gates = 2 x measured on MI355X (depth rel max 2.09e-4, median 3.9e-5; colour abs max 8.9e-5).  This test found rows beyond 865,900
    # reading other rows' features (32-bit row offsets in the fused kernels' operand DMA: every no_grad chunk of more than 1,691 rays at
    # N = 512): depth rel 3.5e-2 on the chunk's later rays, 5e-5 on its first ones
    assert float(rel.max()) < 4.5e-4 and float(rel.median()) < 8e-5, (float(rel.max()), float(rel.median()))
    assert float(cerr.max()) < 2e-4, float(cerr.max())


@pytest.mark.gpu
def test_converted_maps_are_reused_for_the_same_tensors_and_follow_their_version_counters():
    """SceneRF.cache_converted_maps (round 6): the S source frames of one image hand the SAME (C,H,W) map tensors to render_rays_batch
    (scenerf.py:154-156) -- their (H,W,C) copies are made once and reused, keyed by tensor identity and version counter: a second call
    on the same tensors reuses the buffers (same result), an in-place edit of a map is seen by the next call (same result as a model
    without the cache)."""
    from scenerf_amd import synth
    kw = dict(sphere_W=376, sphere_H=114, n_pts_uni=32, n_pts_per_gaussian=8)
    R = 64
    mlp, mlpg = synth.mlp_state(41, 4), synth.mlp_state(42, 2, out_scale=4.0)
    maps = {k: v.to(DEV) for k, v in synth.feature_maps(376, 114, 43, smooth=False).items()}
    pix = synth.stride2_pixels((1220, 370), R, 44).to(DEV)
    nu, ng = synth.sampling_noise(R, 32, 32, 45)
    K, T = synth.kitti_cam_K().to(DEV), synth.rel_pose(2.0, 10.0).to(DEV)

    def model(cache):
        m = SceneRF(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8, precision="bf16", **kw).to(DEV)
        m.mlp.load_state_dict(mlp)
        m.mlp_gaussian.load_state_dict(mlpg)
        m.cache_converted_maps = cache
        return m

    def render(m):
        with torch.no_grad():
            return m.render_rays_batch(K, T, maps, sampled_pixels=pix, ray_batch_size=R, noise=(nu.to(DEV), ng.to(DEV)))

    m = model(True)
    o1 = render(m)
    bufs = {i: e[4] for i, e in m._convert_cache.items()}
    assert bufs, "no level was converted (all read in place?)"
    o2 = render(m)
    assert all(m._convert_cache[i][4] is b for i, b in bufs.items())              # the same converted buffers: no second conversion
    assert all(torch.equal(o1[k], o2[k]) for k in OUT_KEYS)
    with torch.no_grad():
        maps["1_1"].mul_(0.5)                                                     # new values in the same tensor: its version moved
    o3 = render(m)
    assert m._convert_cache[0][4] is not bufs[0] or m._convert_cache[0][1] != 0
    ref = render(model(False))
    assert all(torch.equal(o3[k], ref[k]) for k in OUT_KEYS)
    assert not torch.equal(o3["color"], o1["color"])

    # the trained sessions of one image (maps that require grad, autograd on): converted once, each session's map gradients its own
    def trained(m):
        mg = {k: v.clone().requires_grad_(True) for k, v in maps.items()}
        tot = 0.0
        for j in range(2):
            o = m.render_rays_batch(K, T, mg, sampled_pixels=pix, ray_batch_size=R, noise=(nu.to(DEV), ng.to(DEV)))
            if j == 0 and m.cache_converted_maps:
                first = {i: e[4] for i, e in m._convert_cache.items()}
            tot = tot + o["depth"].mean() + o["color"].mean()
        if m.cache_converted_maps:
            assert first and all(m._convert_cache[i][4] is b for i, b in first.items())
        tot.backward()
        return {k: v.grad for k, v in mg.items()}
    ga, gb = trained(model(True)), trained(model(False))
    for k in ga:
        assert float((ga[k] - gb[k]).abs().max()) <= 1e-3 * float(gb[k].abs().max()) + 1e-12, k
