"""Host-side logic that needs no GPU: configuration, boundary (names / signatures / state_dict keys), packing
arithmetic, and the refusal to run the hot path anywhere but on the GPU."""
import inspect
import math

import pytest
import torch

import scenerf_oracle as orc
from scenerf_amd import synth
from scenerf_amd.config import RenderConfig
from scenerf_amd.model import SceneRF, SceneRFBundleFusion
from scenerf_amd.renderer import MLP_PARAM_NAMES, OUTPUT_KEYS


def test_config_matches_oracle_constants():
    for mk_r, mk_o in ((RenderConfig.kitti, orc.OracleConfig.kitti), (RenderConfig.bundlefusion, orc.OracleConfig.bundlefusion)):
        r, o = mk_r(), mk_o()
        assert r.fov == pytest.approx(o.fov)
        assert r.n_samples == o.n_samples
        assert (r.gauss_floor, r.kl_std_floor, r.max_sample_depth, r.std, r.som_sigma) == (
            o.gauss_floor, o.kl_std_floor, o.max_sample_depth, o.std, o.som_sigma)
    c = RenderConfig.kitti(n_pts_uni=64, n_pts_per_gaussian=16).to_c()
    assert c.n_samples == 128 and c.uni_step == pytest.approx((100 - 0.2) / 64)
    # Q1: divisor is W//s but the map is round(W/s): they differ at scales 8 and 16
    assert list(c.map_W) == [1500, 750, 375, 188, 94] and list(c.div_W) == [1500, 750, 375, 187, 93]
    assert list(c.map_H) == [452, 226, 113, 56, 28] and list(c.div_H) == [452, 226, 113, 56, 28]
    assert list(c.map_C) == [80, 160, 320, 640, 1280]


def test_config_validation():
    with pytest.raises(ValueError):
        RenderConfig.kitti(n_gaussians=9).validate()
    with pytest.raises(ValueError):
        RenderConfig.kitti(n_pts_uni=512, n_pts_per_gaussian=8).validate()   # 544 samples > 512
    with pytest.raises(ValueError):
        RenderConfig.kitti(precision="fp16").validate()


# reference state_dict keys of mlp / mlp_gaussian / pe (SURVEY §5 checkpoint row; resnetfc.py:88-118, pe.py:22-30)
REF_KEYS = ["pe._freqs", "pe._phases"]
for _m in ("mlp", "mlp_gaussian"):
    REF_KEYS += ["%s.lin_in.weight" % _m, "%s.lin_in.bias" % _m, "%s.lin_out.weight" % _m, "%s.lin_out.bias" % _m]
    for _b in range(3):
        REF_KEYS += ["%s.blocks.%d.fc_0.weight" % (_m, _b), "%s.blocks.%d.fc_0.bias" % (_m, _b),
                     "%s.blocks.%d.fc_1.weight" % (_m, _b), "%s.blocks.%d.fc_1.bias" % (_m, _b),
                     "%s.lin_z.%d.weight" % (_m, _b), "%s.lin_z.%d.bias" % (_m, _b)]


def test_boundary_state_dict_keys_and_init():
    m = SceneRF(som_sigma=2.0)
    assert sorted(m.state_dict().keys()) == sorted(REF_KEYS)
    assert sum(p.numel() for p in m.mlp.parameters()) == 5410820
    assert sum(p.numel() for p in m.mlp_gaussian.parameters()) == 5409794
    # reference init: fc_1 zero, biases zero (resnetfc.py:33-40)
    assert float(m.mlp.blocks[0].fc_1.weight.abs().max()) == 0.0
    assert float(m.mlp.lin_in.bias.abs().max()) == 0.0
    # pe buffers: f_k = pi 2^k repeated twice, phases 0 / pi/2
    assert m.pe._freqs.flatten().tolist() == pytest.approx([math.pi * 2 ** (k // 2) for k in range(12)], rel=1e-6)
    assert [n for n, _ in m.mlp.named_parameters()] and set(MLP_PARAM_NAMES) == {n for n, _ in m.mlp.named_parameters()}


def test_boundary_signatures_match_reference():
    sig = inspect.signature(SceneRF.__init__)
    ref = ["self", "som_sigma", "lr", "weight_decay", "img_size", "n_rays", "max_infer_depth", "max_sample_depth", "eval_depth",
           "std", "n_gaussians", "n_pts_uni", "n_pts_per_gaussian", "sampling_method", "batch_size", "add_fov_hor", "add_fov_ver",
           "sphere_H", "sphere_W", "use_color", "use_reprojection"]
    assert list(sig.parameters)[:len(ref)] == ref                         # scenerf.py:23-43
    d = {k: v.default for k, v in sig.parameters.items()}
    assert (d["lr"], d["n_rays"], d["std"], d["n_pts_uni"], d["sphere_H"], d["sphere_W"]) == (1e-5, 1200, 2.5, 32, 452, 1500)
    r = inspect.signature(SceneRF.render_rays_batch)
    assert list(r.parameters)[:8] == ["self", "cam_K", "T_source2infer", "x_rgb", "depth_window", "T_cam2velo", "sampled_pixels",
                                      "ray_batch_size"]                  # scenerf.py:392-399
    assert r.parameters["ray_batch_size"].default == 128 and r.parameters["depth_window"].default == 100
    assert OUTPUT_KEYS == ["depth", "color", "gaussian_means", "gaussian_stds", "weights_at_depth", "closest_pts_to_depths",
                           "loss_kl", "alphas", "som_vars", "densities", "weights", "depth_volumes"]   # scenerf.py:456-469
    b = SceneRFBundleFusion(som_sigma=0.02)
    assert b.render_cfg.gauss_floor == 0.5 and b.render_cfg.fov == pytest.approx(orc.OracleConfig(
        v_angle_max=112.2911, v_angle_min=67.6248, h_angle_max=118.6861, h_angle_min=61.2383).fov)


def test_reference_state_dict_loads():
    m = SceneRF(som_sigma=2.0)
    m.mlp.load_state_dict(synth.mlp_state(1, 4))
    m.mlp_gaussian.load_state_dict(synth.mlp_state(2, 2))
    assert [tuple(p.shape) for p in m.mlp.ordered_params()][:4] == [(512, 42), (512,), (4, 512), (4,)]


def test_load_from_checkpoint_reads_a_lightning_format_file(tmp_path):
    """SceneRF.load_from_checkpoint as the reference's evaluation / reconstruction scripts call it (render_colors.py:38-40,
    save_depth_metrics.py:57, generate_novel_depths.py:48, *_bf.py:58-59), on a Lightning-format file written with the reference's key
    names: {"state_dict": <REF_KEYS + the encoder under net_rgb.>, "hyper_parameters": <the reference ctor's arguments>}.  Works with or
    without pytorch_lightning installed (the fallback base class implements it)."""
    import warnings
    src = SceneRF(som_sigma=1.25, n_pts_uni=64, n_pts_per_gaussian=16, std=2.0, add_fov_hor=20, add_fov_ver=8)
    src.mlp.load_state_dict(synth.mlp_state(5, 4))
    src.mlp_gaussian.load_state_dict(synth.mlp_state(6, 2))
    sd = {k: v.clone() for k, v in src.state_dict().items()}
    assert sorted(sd) == sorted(REF_KEYS)
    sd["net_rgb.encoder.conv_stem.weight"] = torch.randn(8, 3, 3, 3)        # the reference's encoder travels in the same file
    sd["net_rgb.decoder.conv2.bias"] = torch.randn(8)
    hp = dict(som_sigma=1.25, lr=1e-5, weight_decay=0, img_size=(1220, 370), n_rays=1200, max_infer_depth=120, max_sample_depth=100,
              eval_depth=80, std=2.0, n_gaussians=4, n_pts_uni=64, n_pts_per_gaussian=16, sampling_method="uniform", batch_size=1,
              add_fov_hor=20, add_fov_ver=8, sphere_H=452, sphere_W=1500, use_color=True, use_reprojection=True)   # scenerf.py:23-43
    path = str(tmp_path / "last.ckpt")
    torch.save({"state_dict": sd, "hyper_parameters": hp, "epoch": 3, "pytorch-lightning_version": "1.4.9"}, path)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m = SceneRF.load_from_checkpoint(path)
    assert isinstance(m, SceneRF)
    assert (m.n_pts_uni, m.n_pts_per_gaussian, m.std, float(m.ray_som.som_sigma)) == (64, 16, 2.0, 1.25)
    assert m.render_cfg.n_samples == 128 and m.render_cfg.add_fov_hor == 20
    for k, v in src.state_dict().items():
        assert torch.equal(m.state_dict()[k], v), k
    if hasattr(m, "_skipped_checkpoint_keys"):      # (the no-Lightning base class: says which encoder tensors it left out)
        assert m._skipped_checkpoint_keys == ["net_rgb.encoder.conv_stem.weight", "net_rgb.decoder.conv2.bias"]
        assert any("net_rgb" in str(x.message) for x in w)
        # keyword overrides like Lightning's, and an injected encoder takes its tensors
        enc = torch.nn.Module()
        enc.encoder = torch.nn.Module(); enc.encoder.conv_stem = torch.nn.Conv2d(3, 8, 3, bias=False)
        enc.decoder = torch.nn.Module(); enc.decoder.conv2 = torch.nn.Conv2d(8, 8, 1)
        sd["net_rgb.decoder.conv2.weight"] = torch.randn(8, 8, 1, 1)
        torch.save({"state_dict": sd, "hyper_parameters": hp}, path)
        m2 = SceneRF.load_from_checkpoint(path, net_rgb=enc, precision="bf16")
        assert m2.render_cfg.precision == "bf16" and torch.equal(m2.net_rgb.encoder.conv_stem.weight, sd["net_rgb.encoder.conv_stem.weight"])
        with pytest.raises(RuntimeError, match="missing keys"):
            bad = {k: v for k, v in sd.items() if k != "mlp.lin_in.weight"}
            torch.save({"state_dict": bad, "hyper_parameters": hp}, path)
            SceneRF.load_from_checkpoint(path)
    b = SceneRFBundleFusion(som_sigma=0.02, sample_grid_size=3)
    torch.save({"state_dict": b.state_dict(), "hyper_parameters": dict(b.hparams)}, path)
    b2 = SceneRFBundleFusion.load_from_checkpoint(path)
    assert b2.sample_grid_size == 3 and b2.render_cfg.gauss_floor == 0.5


def test_hot_path_refuses_cpu_tensors():
    """No CPU / eager fallback: CPU inputs raise instead of silently running something else."""
    m = SceneRF(som_sigma=2.0, sphere_W=376, sphere_H=114)
    maps = {k: torch.zeros(s) for k, s in synth.feature_map_shapes(376, 114).items()}
    with pytest.raises(RuntimeError, match="GPU"):
        m.render_rays_batch(synth.kitti_cam_K(), torch.eye(4), maps, sampled_pixels=torch.zeros(8, 2), ray_batch_size=8)
    with pytest.raises(RuntimeError, match="HIP MLP pass"):
        m.mlp(torch.zeros(1, 2522))
    # empty / missing ray sets fail up front with a clear message (the reference dies in torch.cat over zero chunks)
    with pytest.raises(ValueError, match="empty"):
        m.render_rays_batch(synth.kitti_cam_K(), torch.eye(4), maps, sampled_pixels=torch.zeros(0, 2), ray_batch_size=8)
    with pytest.raises(ValueError, match="required"):
        m.render_rays_batch(synth.kitti_cam_K(), torch.eye(4), maps)


def test_channels_last_maps_are_recognised_and_carried_in_the_call_config():
    """The in-place channels-last entry (renderer.HWC / a (C,H,W) tensor with (H,W,C) memory): detection and the per-call
    scenerf_cfg.map_chw state are host logic -- checked here without a GPU (the session stops at the first device requirement)."""
    from scenerf_amd.config import RenderConfig
    from scenerf_amd.renderer import HWC, RenderSession
    cfg = RenderConfig.kitti(sphere_W=376, sphere_H=114)
    assert list(cfg.to_c().map_chw) == [0, 0, 0, 1, 1]                        # (3, 4): read from the caller's (C,H,W) tensor
    import dataclasses
    c2 = dataclasses.replace(cfg, hwc_scales=(0, 1, 3), direct_scales=(4,))
    assert list(c2.to_c().map_chw) == [2, 2, 0, 2, 1]
    with pytest.raises(ValueError, match="HWC expects"):
        HWC(torch.zeros(2, 3, 4, 5))
    w = HWC(torch.zeros(5, 7, 3))
    assert w.shape == (5, 7, 3) and w.device.type == "cpu" and w.detach().t.shape == (5, 7, 3)
    shapes = synth.feature_map_shapes(376, 114)
    maps = {}
    for k, (c, h, wd) in shapes.items():
        maps[k] = torch.empty_strided((c, h, wd), (1, wd * c, c))             # what a torch.channels_last batch slice looks like
    b4 = torch.zeros(1, *shapes["1_1"]).to(memory_format=torch.channels_last)[0]
    assert b4.stride() == maps["1_1"].stride()
    with pytest.raises(RuntimeError, match="GPU"):                            # recognised as channels-last, then refused for being on the CPU
        RenderSession(cfg, maps, [], [])
    with pytest.raises(RuntimeError, match="GPU"):
        RenderSession(cfg, {k: HWC(v.permute(1, 2, 0)) for k, v in maps.items()}, [], [])


def test_split_bf16_linin_arithmetic():
    """The bf16-mode lin_in split (x_hi.w_hi + x_lo.w_hi + x_hi.w_lo) reproduces fp32 to ~2^-15 relative."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(256, 48, generator=g) * 40.0
    w = torch.randn(512, 48, generator=g) * 0.02
    xh = x.to(torch.bfloat16).float()
    xl = (x - xh).to(torch.bfloat16).float()
    wh = w.to(torch.bfloat16).float()
    wl = (w - wh).to(torch.bfloat16).float()
    got = xh @ wh.T + xl @ wh.T + xh @ wl.T
    ref = x.double() @ w.double().T
    plain = xh @ wh.T
    scale = (x.abs().double() @ w.abs().double().T).max()
    assert float((got.double() - ref).abs().max() / scale) < 5e-5
    assert float((plain.double() - ref).abs().max() / scale) > 5e-4     # what the split buys


def test_synthetic_inputs_are_deterministic():
    a, b = synth.feature_maps(64, 24, 5), synth.feature_maps(64, 24, 5)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert [tuple(v.shape) for v in a.values()] == [(80, 24, 64), (160, 12, 32), (320, 6, 16), (640, 3, 8), (1280, 2, 4)]
    p = synth.stride2_pixels((1220, 370), 100, 3)
    assert p.shape == (100, 2) and float(p[:, 0].max()) <= 1218 and bool(((p % 2) == 0).all())
    assert len({(float(u), float(v)) for u, v in p}) == 100


def test_spherical_mapping_from_pixels_matches_reference_formula_and_caches_the_grid():
    """SphericalMapping.from_pixels (encoder-side, spherical_mapping.py:80-115): full-image call equals the explicit-coordinates call,
    and the pixel grid is built once."""
    import math
    import torch
    from scenerf_amd import synth
    from scenerf_amd.model import SceneRF
    m = SceneRF(som_sigma=2.0, add_fov_hor=20, add_fov_ver=8, img_size=(61, 19), sphere_W=75, sphere_H=23)
    K = synth.kitti_cam_K().clone()
    K[0, 0] = K[1, 1] = 36.0; K[0, 2] = 30.0; K[1, 2] = 9.0
    inv_K = torch.inverse(K)
    pix, sph, dist = m.spherical_mapping.from_pixels(inv_K=inv_K)
    assert pix.shape == (61 * 19, 2) and sph.dtype == torch.long
    grid_a = m.spherical_mapping._full_grid(inv_K)
    pix2, sph2, dist2 = m.spherical_mapping.from_pixels(inv_K=inv_K)
    assert m.spherical_mapping._full_grid(inv_K) is grid_a            # cached
    assert torch.equal(sph, sph2) and torch.equal(dist, dist2)
    # explicit coordinates, straight from the reference's formulas
    ys, xs = torch.meshgrid(torch.arange(19), torch.arange(61), indexing="ij")
    pc = torch.stack([xs.reshape(-1), ys.reshape(-1)], 1).float()
    _, sph3, _ = m.spherical_mapping.from_pixels(inv_K=inv_K, pix_coords=pc)
    assert torch.equal(sph, sph3)
    cam = (inv_K @ torch.cat([pc, torch.ones(len(pc), 1)], 1).T).T
    d = cam.norm(dim=1)
    v = torch.acos(-cam[:, 1] / d) / math.pi * 180
    sm = m.spherical_mapping
    py = torch.round((v - sm.v_angle_min) / sm.v_fov * (sm.out_img_H - 1)).long()
    assert torch.equal(sph[:, 1], py)


def test_uniform_only_branch_configuration():
    """scenerf.py:647-650 / scenerf_bf.py:662-665: n_pts_uni == 0 and n_pts_per_gaussian == 1 renders the uniform samples alone.  The
    BundleFusion model substitutes 2 uniform samples (scenerf_bf.py:623-626): N = 2, the C config carries the uniform-only flag.  The
    KITTI model has no substitute: the reference divides by zero in uniform_sampling (utils.py:77) -- same exception type here."""
    import pytest
    from scenerf_amd import _capi
    from scenerf_amd.config import RenderConfig
    with pytest.raises(ZeroDivisionError, match="division by zero"):
        RenderConfig.kitti(n_pts_uni=0, n_pts_per_gaussian=1).validate()
    RenderConfig.kitti(n_pts_uni=0, n_pts_per_gaussian=2).validate()      # gaussians only: the `else` branch (scenerf.py:651-654)
    bf = RenderConfig.bundlefusion(n_pts_uni=0, n_pts_per_gaussian=1)
    assert bf.uniform_only and bf.n_samples == 2 and bf.n_uni_used == 2 and bf.n_uni_drawn == 2
    c = bf.to_c()
    assert c.n_samples == 2 and c.n_pts_uni == 2 and c.n_pts_per_gaussian == 1 and (c.flags & _capi.FLAG_UNIFORM_ONLY)
    assert abs(c.uni_step - (12.0 - 0.2) / 2) < 1e-6
    go = RenderConfig.bundlefusion(n_pts_uni=0, n_pts_per_gaussian=8)      # gaussian-only: the 2 substitutes are drawn and discarded
    assert not go.uniform_only and go.n_samples == 32 and go.n_uni_used == 0 and go.n_uni_drawn == 2
    assert not (go.to_c().flags & _capi.FLAG_UNIFORM_ONLY) and go.to_c().n_pts_uni == 0


def test_pixel_grid_follows_the_reference_scripts():
    """render_colors.py:102-111 / generate_novel_depths.py:103-112: meshgrid(xs, ys) ('ij'), cat on the last axis, reshape(-1, 2)."""
    from scenerf_amd.inference import pixel_grid
    for stride in (1, 2, 3):
        xs = torch.arange(start=0, end=1220, step=stride).float()
        ys = torch.arange(start=0, end=370, step=stride).float()
        gx, gy = torch.meshgrid(xs, ys, indexing="ij")
        want = torch.cat([gx.unsqueeze(-1), gy.unsqueeze(-1)], dim=2).reshape(-1, 2)
        assert torch.equal(pixel_grid((1220, 370), stride, "cpu"), want)
    assert pixel_grid((1220, 370), 1, "cpu").shape[0] == 451400 and pixel_grid((1220, 370), 3, "cpu").shape[0] == 407 * 124


def test_parity_regression_gate_fires_between_1_2_and_1_3_times_the_measured_values():
    """tests/test_gpu_parity_full.py holds every case to 1.25 x the errors measured for the committed kernels
    (tests/golden/parity_full_measured.json).  The gate itself, on a made-up report: 1.2 x everything passes, 1.3 x any one of
    an output, a gradient group, the loss or the head's offsets is reported; values at rounding noise sit under the floors; a case
    without a stored reference is not gated."""
    import copy
    import json
    import os
    import test_gpu_parity_full as pf
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parity_full_measured.json")
    measured = json.load(open(path))
    assert {"kitti_c2_r1200_n128_bf16", "kitti_c2_r1200_n128_bf16_hwc", "kitti_default_r1200_n64_bf16", "kitti_default_r1200_n64_fp32",
            "bf_c4_r1080_n96_bf16", "kitti_c5_r32_n512_bf16"} <= set(measured)
    key = "kitti_c2_r1200_n128_bf16"
    ref = measured[key]

    def report(scale, bump=None):
        out = {k: {"max_abs": v * scale, "max_rel": v * scale} for k, v in ref["out"].items()}
        grad = {g + "w": {"rel_l2": v * scale} for g, v in ref["grad"].items()}
        rep = {"matched": {"out": out, "grad": grad, "loss": {"rel": ref["loss"] * scale}}, "head_offsets": {"rel_l2": ref["head"] * scale}}
        if bump == "out":
            rep["matched"]["out"]["depth"] = {"max_abs": 1.0, "max_rel": ref["out"]["depth"] * 1.3}
        elif bump == "grad":
            rep["matched"]["grad"]["mlp.w"]["rel_l2"] = ref["grad"]["mlp."] * 1.3
        elif bump == "loss":
            rep["matched"]["loss"]["rel"] = ref["loss"] * 1.3
        elif bump == "head":
            rep["head_offsets"]["rel_l2"] = ref["head"] * 1.3
        return rep

    assert pf._regression_fails(key, report(1.2)) == []
    assert len(pf._regression_fails(key, report(1.3))) >= 4
    for what in ("out", "grad", "loss", "head"):
        fails = pf._regression_fails(key, report(1.0, bump=what))
        assert len(fails) == 1 and "[regression]" in fails[0], (what, fails)
    assert pf._regression_fails("no_such_case_bf16", report(100.0)) == []
    # fp32 values at rounding noise: the floors, not 1.25 x 3e-8
    k32 = "kitti_c2_r1200_n128_fp32"
    r32 = copy.deepcopy(measured[k32])
    rep = {"matched": {"out": {k: {"max_abs": 1.5e-6, "max_rel": 1.5e-6} for k in r32["out"]}, "grad": {}, "loss": {"rel": 1.5e-6}},
           "head_offsets": {"rel_l2": 9e-7}}
    assert [f for f in pf._regression_fails(k32, rep) if "output" in f and "depth_volumes" in f] == []


def test_more_hw_queues_respects_the_user_and_the_initialised_runtime(monkeypatch):
    """scenerf_amd.dist.more_hw_queues: GPU_MAX_HW_QUEUES = 8 for ranks of a process group, only if the user has not set it and the HIP
    runtime has not initialised yet (the variable is read once, at initialisation)."""
    import os
    from scenerf_amd import dist as sdist
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    assert sdist.more_hw_queues() is True and os.environ["GPU_MAX_HW_QUEUES"] == "8"
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "5")
    assert sdist.more_hw_queues() is True and os.environ["GPU_MAX_HW_QUEUES"] == "5"
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "is_initialized", lambda: True)
    assert sdist.more_hw_queues() is False and "GPU_MAX_HW_QUEUES" not in os.environ


def test_resnetfc_container_is_generic_like_the_reference():
    """resnetfc.py:67-131 is generic in n_blocks / d_hidden; the container mirrors that (state_dict names, shapes, init) and says which
    shapes the fused kernels take (``is_standard``); what the ray pipeline fixes (d_in = 42, d_latent = 2480, d_out in {2, 4}) is refused."""
    from scenerf_amd.model import ResnetFC
    std = ResnetFC()
    assert std.is_standard and len(std.ordered_params()) == 22
    small = ResnetFC(d_in=42, d_out=4, n_blocks=1, d_hidden=128)
    assert not small.is_standard
    sd = small.state_dict()
    assert sd["lin_in.weight"].shape == (128, 42) and sd["lin_z.0.weight"].shape == (128, 2480) and sd["lin_out.weight"].shape == (4, 128)
    assert "blocks.1.fc_0.weight" not in sd and float(sd["blocks.0.fc_1.weight"].abs().max()) == 0.0      # fc_1 starts at zero (resnetfc.py:40)
    assert [tuple(p.shape) for p in small.ordered_params()] == [(128, 42), (128,), (4, 128), (4,), (128, 128), (128,), (128, 128), (128,),
                                                                (128, 2480), (128,)]
    for bad in (dict(d_in=3), dict(d_latent=512), dict(d_out=3), dict(d_hidden=100), dict(n_blocks=0)):
        with pytest.raises(ValueError):
            ResnetFC(**bad)


def test_bench_reads_each_roofline_from_its_own_counter_file():
    """bench.py quotes PMC bytes from the committed rocprofv3 passes: the step's kernels from profiles/r*_pmc_hbm.json, the per-ray tail from
    r*_tail_pmc_hbm.json -- two files matching one glob (round 5: the tail's file shadowed the step's and `roofline.traffic` came out null)."""
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    fwd = bench.pmc_traffic("mlp_fwd_fused", 153600)
    wg = bench.pmc_traffic("gemm_wgrad_fc", 153600)
    assert fwd and fwd["kernel_fn"].startswith("mlp_wide_kernel<0>") and "_tail_" not in fwd["source"] and fwd["bytes_per_launch"] > 1e9
    assert wg and wg["kernel_fn"].startswith("wgrad_tr_kernel") and wg["bytes_per_launch"] > 1e9
    tail = bench._tail_traffic(128)
    assert tail and "_tail_" in tail["source"] and 1.0 <= tail["tail_fwd"]["ratio"] <= 1.3 and 1.0 <= tail["tail_bwd"]["ratio"] <= 1.3


def test_roofline_min_names_the_worst_big_mfma_bound_kernel_and_the_useful_fraction_excludes_padding():
    """bench.py::_roofline_extras: `roofline` stays the LONGEST kernel; `roofline_min` = the lowest fraction of the
    MFMA peak among the MFMA-bound kernels above 10 % of the step -- small kernels and the atomics-bound feature-map scatter do not
    compete, and two kernels trading the longest place do not move it; `roofline_useful_frac` prices the dominant kernel without the
    lin_in problem's padded columns."""
    import os
    import sys
    import types
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    rows = 153600
    mk = lambda name, ms, tflops: {"name": name, "total_ms": ms, "launches": 1, "flops": tflops * 1e12 * ms * 1e-3}      # noqa: E731
    ks = [mk("mlp_fwd_fused", 0.60, 940.0), mk("mlp_bwd_fused", 0.49, 1000.0), mk("gemm_wgrad_fc", 0.66, 970.0),
          mk("gemm_dfeat_scatter", 0.24, 250.0), mk("mlp_fwd_fused/g", 0.12, 170.0)]
    args = types.SimpleNamespace(step_ms_for_roofline=2.35, rows_per_launch=rows)
    out = bench._roofline_extras(ks, 2500.0, 1, args)
    assert out["roofline_min_kernel"] == "mlp_fwd_fused" and abs(out["roofline_min_frac"] - 0.376) < 1e-3
    assert set(out["roofline_min"]["candidates"]) == {"mlp_fwd_fused", "mlp_bwd_fused", "gemm_wgrad_fc"}
    # the wgrad and forward kernels trade the longest place: the selection does not move
    ks[0]["total_ms"], ks[0]["flops"] = 0.67, 940.0e12 * 0.67e-3
    assert bench._roofline_extras(ks, 2500.0, 1, args)["roofline_min_kernel"] == "mlp_fwd_fused"
    # useful fraction of the dominant kernel: its FLOPs minus lin_in's 214 padded columns (2 x rows x 512 x 214)
    ks[0]["total_ms"], ks[0]["flops"] = 0.60, 940.0e12 * 0.60e-3
    pad = 2.0 * rows * 512 * (256 - 42)
    want = (ks[2]["flops"] - pad) / 0.66e-3 / 1e12 / 2500.0
    assert abs(out["roofline_useful_frac"] - want) < 1e-4 and out["roofline_useful_frac"] < 970.0 / 2500.0


def test_the_trainers_per_image_scope_ends_with_forward_whatever_happens():
    """training.TrainingMixin._params_fixed: the shared sessions, the packed-operand cache and the pre-drawn pixel subsets exist only while
    ``forward`` runs -- a render_rays_batch call after it (validation, another optimizer step in between) must never see them."""
    m = SceneRF(som_sigma=2.0, img_size=(64, 48), n_rays=16, sphere_H=48, sphere_W=64)
    keys = ("_pack_cache", "_image_sessions", "_predrawn_idx")
    with m._params_fixed():
        assert "_pack_cache" in m.__dict__ and "_image_sessions" in m.__dict__
        m.__dict__["_predrawn_idx"] = [torch.zeros(1)]
    assert not any(k in m.__dict__ for k in keys)
    with pytest.raises(RuntimeError):
        with m._params_fixed():
            raise RuntimeError("a step that fails half-way")
    assert not any(k in m.__dict__ for k in keys)
    assert m.share_image_sessions and m.cache_converted_maps and m.overlap_metric_renders and not m.metric_stream_low_priority
