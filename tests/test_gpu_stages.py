"""Per-kernel parity on the GPU: every stage of the hot path is driven through the C ABI (ctypes) with the
oracle's intermediates as inputs and compared with the oracle's outputs of the same stage.

Index outputs (sort permutation, closest-sample index, BMU-derived masks) are compared exactly given identical
inputs.  So are the spherical indices (round 5): ray directions, sample points, projected pixels and both angles follow torch-CPU's
operation sequence (csrc/sphere_exact.h), with acos pinned to SLEEF u10 because torch.acos itself is not one function
(oracle/sleef_acos.py) -- every index equals the oracle's under that rule, with no window around the .5 boundaries.
"""
import ctypes as C

import pytest
import torch

import scenerf_oracle as orc
from golden_util import Golden
from scenerf_amd import _capi, synth
from scenerf_amd.config import RenderConfig

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _st():
    return torch.cuda.current_stream().cuda_stream


def _cfgs(g: Golden, precision="fp32"):
    ocfg = (orc.OracleConfig.kitti if g.variant == "kitti" else orc.OracleConfig.bundlefusion)(**g.cfg_kwargs())
    rcfg = (RenderConfig.kitti if g.variant == "kitti" else RenderConfig.bundlefusion)(precision=precision, **g.cfg_kwargs())
    return ocfg, rcfg


@pytest.fixture(scope="module")
def case():
    """Oracle run with intermediates for one golden case (small sphere so the maps are cheap)."""
    import dataclasses
    g = Golden("kitti_small_n64")
    ocfg, _ = _cfgs(g)
    ocfg = dataclasses.replace(ocfg, index_rule="pinned")     # the geometry of the path, host-independent (OracleConfig.index_rule)
    mlp, mlpg = g.mlp_states()
    maps = g.feature_maps()
    out = orc.render_chunk(ocfg, mlp, mlpg, g.cam_K, g.T, maps, g.pixels, g.noise_u, g.noise_g, keep_intermediates=True)
    return dict(g=g, ocfg=ocfg, mlp=mlp, mlpg=mlpg, maps=maps, out=out)


_KEEP = []


@pytest.fixture(autouse=True)
def _keepalive():
    """Device temporaries handed to the C ABI as raw pointers must outlive the asynchronous kernels."""
    yield
    torch.cuda.synchronize()
    _KEEP.clear()


def dv(t, dtype=None):
    t = t.detach()
    if dtype is not None:
        t = t.to(dtype)
    t = t.contiguous().to(DEV)
    _KEEP.append(t)
    return t


# ------------------------------------------------------------------------------------------------ GEMM building blocks
@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("M,N,K,tile", [(300, 136, 80, 1), (128, 128, 64, 1), (517, 512, 560, 1), (256, 80, 1536, 1),
                                        (517, 512, 560, 2), (130, 512, 48, 2), (1000, 512, 512, 2), (517, 512, 560, 3), (300, 256, 80, 3), (517, 512, 560, 4), (300, 256, 80, 4), (1000, 512, 2992, 4)])
def test_gemm_nt_matches_fp64(prec, M, N, K, tile):
    lib = _capi.load()
    gen = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, generator=gen)
    W = torch.randn(N, K, generator=gen) * 0.1  # asymmetric operands: catches transposed fragments / outputs
    b = torch.randn(N, generator=gen)
    dt = torch.bfloat16 if prec else torch.float32
    Ad, Wd = dv(A, dt), dv(W, dt)
    Cd = torch.full((M, N), float("nan"), device=DEV)
    _capi.check(lib.scenerf_hip_test_gemm_nt(prec, Ad.data_ptr(), Wd.data_ptr(), dv(b).data_ptr(), M, N, K, 1, tile, Cd.data_ptr(), _st()), "gemm_nt")
    ref = torch.relu(Ad.float().double().cpu()) @ Wd.float().double().cpu().T + b.double()
    scale = (torch.relu(Ad.float().cpu()).abs().double() @ Wd.float().cpu().abs().double().T).max()
    err = (Cd.double().cpu() - ref).abs().max()
    assert err <= 2e-6 * scale, "max err %.3e (scale %.3e)" % (err, scale)


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("M,N,K", [(300, 136, 80), (1000, 512, 48), (4099, 256, 160), (33000, 256, 512), (40007, 512, 256)])
def test_gemm_tn_matches_fp64(prec, M, N, K):
    lib = _capi.load()
    gen = torch.Generator().manual_seed(M + N * 5 + K * 11)
    D = torch.randn(M, N, generator=gen) * 0.1
    A = torch.randn(M, K, generator=gen)
    dt = torch.bfloat16 if prec else torch.float32
    Dd, Ad = dv(D, dt), dv(A, dt)
    Cd = torch.zeros((N, K), device=DEV)
    cs = torch.zeros((N,), device=DEV)
    _capi.check(lib.scenerf_hip_test_gemm_tn(prec, Dd.data_ptr(), Ad.data_ptr(), M, N, K, 1, Cd.data_ptr(), cs.data_ptr(), _st()), "gemm_tn")
    cref = Dd.float().double().cpu().sum(0)
    cerr = (cs.double().cpu() - cref).abs().max()
    assert cerr <= 1e-5 * Dd.float().abs().double().cpu().sum(0).max(), "colsum err %.3e" % cerr
    ref = Dd.float().double().cpu().T @ torch.relu(Ad.float().double().cpu())
    scale = (Dd.float().cpu().abs().double().T @ torch.relu(Ad.float().cpu()).abs().double()).max()
    err = (Cd.double().cpu() - ref).abs().max()
    assert err <= 5e-6 * scale, "max err %.3e (scale %.3e)" % (err, scale)


# ------------------------------------------------------------------------------------------------ geometry
def _ray_setup(case, rcfg):
    lib = _capi.load()
    g, o = case["g"], case["out"]
    cc = rcfg.to_c()
    R, U = g.pixels.shape[0], rcfg.n_pts_uni
    iK = torch.inverse(g.cam_K)
    lin = torch.linspace(0.2, rcfg.max_sample_depth, steps=U)
    unit = torch.empty((R, 3), device=DEV)
    vd = torch.empty((R, 3), device=DEV)
    du = torch.empty((R, U), device=DEV)
    _capi.check(lib.scenerf_hip_ray_setup(C.byref(cc), dv(g.pixels).data_ptr(), dv(iK).data_ptr(), dv(g.T).data_ptr(),
                                          dv(lin).data_ptr(), dv(g.noise_u.reshape(R, U)).data_ptr(), None, R, unit.data_ptr(),
                                          vd.data_ptr(), du.data_ptr(), _st()), "ray_setup")
    return unit, vd, du


def test_in_kernel_sampler_noise_statistics_and_call_counter():
    """RenderConfig.device_rng sessions: ray_setup makes the uniform noise of utils.py:84 and gaussian_sample_sort the normal noise of
    utils.py:208-211 themselves (Philox on (element, call); scenerf_hip.h).  Recovered through the outputs they feed: moments of both,
    the call counter advancing once per (ray_setup, gaussian_sample_sort) pair, fresh values per call, the same values from the same
    state."""
    lib = _capi.load()
    rcfg = RenderConfig.kitti(precision="bf16", n_pts_uni=64, n_pts_per_gaussian=16)
    cc = rcfg.to_c()
    R, U, G, P, N = 2048, 64, 4, 16, 128
    gen = torch.Generator().manual_seed(9)
    pix = dv(torch.rand(R, 2, generator=gen) * torch.tensor([1219.0, 369.0]))
    K = synth.kitti_cam_K()
    iK, T = dv(torch.inverse(K).contiguous()), dv(synth.rel_pose(1.0, 0.0))
    lin = dv(torch.linspace(0.2, rcfg.max_sample_depth, steps=U))
    anchors = dv(orc.gaussian_anchor_distances(orc.OracleConfig.kitti(n_pts_uni=U, n_pts_per_gaussian=P)))
    offs = torch.zeros((R, G, 2), device=DEV)
    rng = torch.tensor([777, 0, 0], dtype=torch.int64, device=DEV)

    def chunk():
        unit, vd, du = torch.empty((R, 3), device=DEV), torch.empty((R, 3), device=DEV), torch.empty((R, U), device=DEV)
        _capi.check(lib.scenerf_hip_ray_setup(C.byref(cc), pix.data_ptr(), iK.data_ptr(), T.data_ptr(), lin.data_ptr(), None, rng.data_ptr(), R,
                                              unit.data_ptr(), vd.data_ptr(), du.data_ptr(), _st()), "ray_setup")
        ng = torch.full((R, G * P), float("nan"), device=DEV)
        gm, gs = torch.empty((R, G), device=DEV), torch.empty((R, G), device=DEV)
        ds, zs = torch.empty((R, N), device=DEV), torch.empty((R, N), device=DEV)
        perm = torch.empty((R, N), dtype=torch.int32, device=DEV)
        _capi.check(lib.scenerf_hip_gaussian_sample_sort(C.byref(cc), offs.data_ptr(), anchors.data_ptr(), du.data_ptr(), ng.data_ptr(),
                                                         rng.data_ptr(), unit.data_ptr(), R, gm.data_ptr(), gs.data_ptr(), ds.data_ptr(),
                                                         zs.data_ptr(), perm.data_ptr(), _st()), "gaussian_sample_sort")
        torch.cuda.synchronize()
        nu = ((du - lin[None, :]) / rcfg.to_c().uni_step).double().cpu()
        return nu, ng.double().cpu(), ds.cpu(), gm.cpu(), gs.cpu(), perm.cpu()

    nu1, ng1, ds1, gm1, gs1, perm1 = chunk()
    assert rng.tolist()[:2] == [777, 1]
    nu2, ng2, _, _, _, _ = chunk()
    assert rng.tolist()[:2] == [777, 2]
    for nu in (nu1, nu2):           # U[0, 1): n = 131,072 -> SE of the mean 8e-4
        assert float(nu.min()) >= -1e-4 and float(nu.max()) < 1.0 + 1e-4
        assert abs(float(nu.mean()) - 0.5) < 5e-3 and abs(float(nu.var()) - 1.0 / 12.0) < 2e-3
    for ng in (ng1, ng2):           # N(0, 1): n = 131,072
        assert bool(torch.isfinite(ng).all())
        assert abs(float(ng.mean())) < 1.5e-2 and abs(float(ng.std()) - 1.0) < 1.5e-2
        assert abs(float((ng ** 4).mean()) - 3.0) < 0.15 and abs(float((ng ** 3).mean())) < 0.06
    assert float((nu1 - nu2).abs().mean()) > 0.2 and float((ng1 - ng2).abs().mean()) > 0.5       # fresh per call
    assert abs(float((nu1[:, :-1] * nu1[:, 1:]).mean()) - 0.25) < 5e-3                            # neighbours uncorrelated
    # the sampler used the noise it wrote: samples = clamp(mean + noise * std, 0.1), merged with the uniform ones and sorted
    d_g = torch.clamp(gm1.double().repeat_interleave(P, 1) + ng1 * gs1.double().repeat_interleave(P, 1), min=0.1)
    got = torch.sort(ds1.double(), dim=1).values
    want = torch.sort(torch.cat([(nu1 * rcfg.to_c().uni_step + lin.double().cpu()[None, :]), d_g], dim=1), dim=1).values
    assert float((got - want).abs().max()) < 1e-4
    assert bool((torch.sort(perm1, dim=1).values == torch.arange(N)[None, :]).all())
    rng.copy_(torch.tensor([777, 0, 0]))
    nu1b, ng1b, _, _, _, _ = chunk()
    assert torch.equal(nu1, nu1b) and torch.equal(ng1, ng1b)                                      # same state, same values


def test_ray_setup(case):
    _, rcfg = _cfgs(case["g"])
    unit, vd, du = _ray_setup(case, rcfg)
    o = case["out"]
    # csrc/sphere_exact.h: torch-CPU's operation sequence (matmul as k-ordered fma chains, the 2-norm as fma(z,z,fma(y,y,x*x))) -- bit for bit
    assert torch.equal(unit.cpu(), o["_unit"]), "unit directions must be bit-exact"
    assert torch.equal(vd.cpu(), o["_viewdir"]), "view directions must be bit-exact"
    assert torch.equal(du.cpu(), o["_dist_u"]), "uniform sample distances must be bit-exact"


def _encode(case, rcfg, dist, stride, ppr, M):
    lib = _capi.load()
    g, o = case["g"], case["out"]
    cc = rcfg.to_c()
    iK = torch.inverse(g.cam_K)
    pts = torch.empty((M, 3), device=DEV)
    idx = torch.empty((M, 2), dtype=torch.int32, device=DEV)
    xenc = torch.empty((M, 48), device=DEV)
    x3 = torch.full((M + 1, 144), float("nan"), dtype=torch.bfloat16, device=DEV)     # (one row of slack: scenerf_hip.h, encode_points zero-fills it)
    _capi.check(lib.scenerf_hip_encode_points(C.byref(cc), dist.data_ptr(), stride, ppr, dv(o["_unit"]).data_ptr(),
                                              dv(o["_viewdir"]).data_ptr(), dv(g.cam_K).data_ptr(), dv(iK).data_ptr(),
                                              dv(g.T).data_ptr(), M, pts.data_ptr(), idx.data_ptr(), xenc.data_ptr(), x3.data_ptr(), _st()),
                "encode_points")
    # the split-bf16 form written by the same launch: [hi | lo | hi] with hi = bf16(x), lo = bf16(x - hi), exactly
    hi = xenc.to(torch.bfloat16)
    lo = (xenc - hi.float()).to(torch.bfloat16)
    assert torch.equal(x3[:M], torch.cat([hi, lo, hi], dim=1)), "split encoding differs from bf16 hi / lo of the fp32 encoding"
    assert float(x3[M:].float().abs().max()) == 0.0, "the slack row behind the split encoding must be zero-filled"
    # ... and alone (xenc = NULL: the product's bf16 path)
    x3b = torch.full_like(x3, float("nan"))
    idx2 = torch.empty_like(idx)
    _capi.check(lib.scenerf_hip_encode_points(C.byref(cc), dist.data_ptr(), stride, ppr, dv(o["_unit"]).data_ptr(),
                                              dv(o["_viewdir"]).data_ptr(), dv(g.cam_K).data_ptr(), dv(iK).data_ptr(),
                                              dv(g.T).data_ptr(), M, None, idx2.data_ptr(), None, x3b.data_ptr(), _st()), "encode_points")
    assert torch.equal(x3b, x3) and torch.equal(idx2, idx)
    return pts, idx, xenc


def _check_sphere_idx(idx_gpu, pts_oracle, g, ocfg):
    """sphere indices bit-exact.  Under the pinned rule (oracle ``index_rule="pinned"``: torch's own SLEEF build, which is
    what torch.atan2 is and what torch.acos is without MKL; oracle/sleef_acos.py) EVERY index equals the oracle's -- no window around the
    .5 boundaries, no tolerated flips.  Returned for the log: how many samples the reference's calls as THIS host runs them (torch.acos =
    MKL's vmsAcos, ISA-dependent; K @ p = MKL's sgemm, vendor-dependent) place on a neighbouring texel."""
    import dataclasses
    iK = torch.inverse(g.cam_K)
    idx_rule = orc.sphere_coords(orc.project_to_pixels(pts_oracle, g.cam_K, "pinned"), iK, dataclasses.replace(ocfg, index_rule="pinned"))
    got = idx_gpu.cpu().long()
    # far-out / behind-camera coordinates: the kernel clamps before the int conversion (they are out of every map either way)
    assert torch.equal(got.clamp(-10**9, 10**9), idx_rule.clamp(-10**9, 10**9)), \
        "sphere indices differ from the pinned rule on %d rows" % int((got != idx_rule).any(1).sum())
    idx_host = orc.sphere_coords(orc.project_to_pixels(pts_oracle, g.cam_K, "torch"), iK, dataclasses.replace(ocfg, index_rule="torch"))
    d = got - idx_host
    assert int(d.abs().max()) <= 1
    return int((d != 0).any(1).sum()), 0


def test_device_acos_atan2_are_torchs_sleef():
    """The device executes csrc/sphere_exact.h's sequence: equal to torch's own SLEEF build (oracle/sleef_acos.py) on every input tried,
    special values included.  (That the sequence IS SLEEF's: tests/test_sphere_exact.py, every float32 in [-1, 1], no GPU needed.)"""
    import sleef_acos
    lib = _capi.load()
    gen = torch.Generator().manual_seed(11)
    n = 1 << 21
    sp = torch.tensor([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 1e-30, -1e-30, 1e-40, -1e-40, 3e-39, -3e-39, 0.49999997, 0.50000006, 0.99999994,
                       -0.99999994, float("inf"), float("-inf"), 1e38, -1e38, 2.0, -2.0, 0.70710678, 1e-8])
    x = torch.cat([torch.rand(n, generator=gen) * 2 - 1, sp.clamp(-1, 1), torch.linspace(-1, 1, 100001)])
    y = torch.empty_like(x, device=DEV)
    _capi.check(lib.scenerf_hip_test_acos_atan2(dv(x).data_ptr(), None, x.numel(), y.data_ptr(), None, _st()), "test_acos_atan2")
    assert torch.equal(y.cpu().view(torch.int32), sleef_acos.acos(x).view(torch.int32)), "device acos != torch's SLEEF acosf_u10"
    a = torch.cat([torch.randn(n, generator=gen) * torch.exp(torch.randn(n, generator=gen) * 4), sp.repeat_interleave(sp.numel())])
    b = torch.cat([torch.randn(n, generator=gen) * torch.exp(torch.randn(n, generator=gen) * 4), sp.repeat(sp.numel())])
    y = torch.empty_like(a, device=DEV)
    _capi.check(lib.scenerf_hip_test_acos_atan2(dv(a).data_ptr(), dv(b).data_ptr(), a.numel(), None, y.data_ptr(), _st()), "test_acos_atan2")
    assert torch.equal(y.cpu().view(torch.int32), sleef_acos.atan2(a, b).view(torch.int32)), "device atan2 != torch's SLEEF atan2f_u10"


@pytest.mark.parametrize("variant", ["kitti", "bundlefusion"])
def test_from_pixels_on_the_gpu_follows_the_pinned_rule(variant):
    """SphericalMapping.from_pixels (spherical_mapping.py:80-97) on a CUDA tensor = scenerf_hip_pixels_to_sphere: the full image grid,
    every index equal to the oracle's under the pinned rule, the distance bit-exact."""
    import dataclasses
    from scenerf_amd.model import SceneRF, SceneRFBundleFusion
    if variant == "kitti":
        m = SceneRF(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8)
        ocfg = orc.OracleConfig.kitti(index_rule="pinned")
        K = synth.kitti_cam_K()
    else:
        m = SceneRFBundleFusion(som_sigma=0.02, std=0.1, add_fov_hor=14, add_fov_ver=11, sphere_W=960, sphere_H=720)
        ocfg = orc.OracleConfig.bundlefusion(index_rule="pinned")
        K = torch.tensor([[583.0, 0, 320.0], [0, 583.0, 240.0], [0, 0, 1]])
    iK = torch.inverse(K).contiguous()
    pix, idx, dist = m.spherical_mapping.from_pixels(inv_K=iK.to(DEV))
    W, H = ocfg.img_size
    assert pix.shape == (W * H, 2) and idx.dtype == torch.int64
    ref_idx = orc.sphere_coords(pix.cpu(), iK, ocfg)
    assert torch.equal(idx.cpu(), ref_idx)
    c = orc._matvec(iK, orc._homog(pix.cpu()), "pinned")
    assert torch.equal(dist.cpu(), torch.linalg.norm(c, ord=2, dim=1))


def test_encode_points_main_samples(case):
    g, o, ocfg = case["g"], case["out"], case["ocfg"]
    _, rcfg = _cfgs(g)
    R, N = o["_dist_sorted"].shape
    pts, idx, xenc = _encode(case, rcfg, dv(o["_dist_sorted"]), N, N, R * N)
    assert torch.equal(pts.cpu(), o["_pts_sorted"].reshape(-1, 3)), "sample points must be bit-exact"
    n_diff, _ = _check_sphere_idx(idx, o["_pts_sorted"].reshape(-1, 3), g, ocfg)
    print("sphere idx: equal to the pinned rule on all %d rows; %d rows differ from the reference's calls as this host runs them" % (R * N, n_diff))
    ref = o["_xin"][:, 2480:]
    x = xenc.cpu()
    assert torch.equal(x[:, 42:], torch.zeros(R * N, 6))
    torch.testing.assert_close(x[:, 39:42], ref[:, 39:42], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(x[:, :3], ref[:, :3], rtol=1e-6, atol=1e-5)
    # PE: fp32 argument rounding at |x| f ~ 1e4 rad allows ~1e-3; tight where the argument is small
    arg = (ref[:, :3].abs().max(dim=1).values * 3.1416 * 32)
    small = arg < 100
    torch.testing.assert_close(x[small, 3:39], ref[small, 3:39], rtol=0, atol=2e-5)
    torch.testing.assert_close(x[:, 3:39], ref[:, 3:39], rtol=0, atol=4e-3)


def test_encode_points_anchors(case):
    g, o, ocfg = case["g"], case["out"], case["ocfg"]
    _, rcfg = _cfgs(g)
    R = g.pixels.shape[0]
    G = rcfg.n_gaussians
    anchors = orc.gaussian_anchor_distances(ocfg)
    pts, idx, xenc = _encode(case, rcfg, dv(anchors), 0, G, R * G)
    assert torch.equal(pts.cpu(), o["_anchor_pts"]), "anchor points must be bit-exact"
    _check_sphere_idx(idx, o["_anchor_pts"], g, ocfg)


# ------------------------------------------------------------------------------------------------ gather
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_gather_features(case, precision):
    lib = _capi.load()
    g, o = case["g"], case["out"]
    _, rcfg = _cfgs(g, precision)
    cc = rcfg.to_c()
    prec = rcfg.precision_code
    act = torch.bfloat16 if prec else torch.float32
    hwc = []
    for i, ((c, h, w), key) in enumerate(zip(rcfg.map_shapes(), ["1_1", "1_2", "1_4", "1_8", "1_16"])):
        src = dv(case["maps"][key])
        dst = torch.empty((h, w, c), dtype=act, device=DEV)
        _capi.check(lib.scenerf_hip_maps_chw_to_hwc(src.data_ptr(), dst.data_ptr(), c, h, w, prec, _st()), "maps_chw_to_hwc")
        torch.testing.assert_close(dst.float().cpu(), case["maps"][key].permute(1, 2, 0).to(act).float(), rtol=0, atol=0)
        # RenderConfig.direct_scales: the coarse levels are gathered from the fp32 (C,H,W) tensor itself (scenerf_cfg.map_chw)
        hwc.append(src if i in rcfg.direct_scales else dst)
    idx = o["_idx"].to(torch.int32)
    M = idx.shape[0]
    Mpad = (M + 127) // 128 * 128
    Z = torch.full((Mpad, 2480), float("nan"), dtype=act, device=DEV)
    mask = torch.zeros((Mpad // 128,), dtype=torch.uint8, device=DEV)
    tex = torch.empty((M, 5, 4), dtype=torch.int32, device=DEV)
    tw = torch.empty((M, 5, 4), device=DEV)
    arr = (C.c_void_p * 5)(*[t.data_ptr() for t in hwc])
    _capi.check(lib.scenerf_hip_gather_features(C.byref(cc), C.byref(arr), dv(idx).data_ptr(), M, Z.data_ptr(), mask.data_ptr(),
                                                tex.data_ptr(), tw.data_ptr(), _st()), "gather_features")
    ref = o["_xin"][:, :2480]
    Zc, mk = Z.float().cpu(), mask.cpu()
    off = 0
    for s, (c, h, w) in enumerate(rcfg.map_shapes()):
        for t in range(Mpad // 128):
            rows = slice(t * 128, min((t + 1) * 128, M))
            r = ref[rows, off:off + c]
            if (mk[t] >> s) & 1:
                tol = 1e-5 if prec == 0 else 1.2e-2
                torch.testing.assert_close(Zc[rows, off:off + c], r, rtol=tol, atol=tol)
            else:
                assert float(r.abs().max()) == 0.0, "tile %d scale %d skipped but the oracle has non-zero features" % (t, s)
                # columns below SCENERF_Z_DENSE_COLS are always defined (exact zeros: the batched lin_z weight gradient reads them for
                # every row); beyond that an untouched (tile, scale) pair is left unwritten (the NaN fill survives)
                dense = max(0, min(off + c, 256) - off)
                full = slice(t * 128, (t + 1) * 128)
                if dense:
                    assert float(Zc[full, off:off + dense].abs().max()) == 0.0
                if dense < c:
                    assert bool(torch.isnan(Zc[full, off + dense:off + c]).all())
        off += c
    # taps are a faithful description of the gather: re-applying them on the CHW map reproduces the oracle
    key = "1_1"
    fm = case["maps"][key].reshape(80, -1)
    texc, twc = tex[:, 0].cpu().long(), tw[:, 0].cpu()
    rec = torch.zeros(M, 80)
    for t in range(4):
        ok = texc[:, t] >= 0
        rec[ok] += (fm[:, texc[ok, t]] * twc[ok, t]).T
    torch.testing.assert_close(rec, ref[:, :80], rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------------------------------------ sampler + sort
def test_gaussian_sample_sort(case):
    lib = _capi.load()
    g, o, ocfg = case["g"], case["out"], case["ocfg"]
    _, rcfg = _cfgs(g)
    cc = rcfg.to_c()
    R, N, G = g.pixels.shape[0], rcfg.n_samples, rcfg.n_gaussians
    anchors = orc.gaussian_anchor_distances(ocfg)
    gm = torch.empty((R, G), device=DEV)
    gs = torch.empty((R, G), device=DEV)
    ds = torch.empty((R, N), device=DEV)
    zs = torch.empty((R, N), device=DEV)
    perm = torch.empty((R, N), dtype=torch.int32, device=DEV)
    _capi.check(lib.scenerf_hip_gaussian_sample_sort(C.byref(cc), dv(o["_offsets"]).data_ptr(), dv(anchors).data_ptr(),
                                                     dv(o["_dist_u"]).data_ptr(), dv(g.noise_g).data_ptr(), None, dv(o["_unit"]).data_ptr(),
                                                     R, gm.data_ptr(), gs.data_ptr(), ds.data_ptr(), zs.data_ptr(), perm.data_ptr(),
                                                     _st()), "gaussian_sample_sort")
    assert torch.equal(gm.cpu(), o["gaussian_means"].detach())
    assert torch.equal(gs.cpu(), o["gaussian_stds"].detach())
    assert torch.equal(ds.cpu(), o["_dist_sorted"].detach()), "sorted distances must be bit-exact"
    assert torch.equal(zs.cpu(), o["depth_volumes"].detach())
    # permutation: bit-exact where keys are unique (ties = identical clamped samples, order irrelevant; torch's
    # CPU argsort is not stable there)
    p_ref = o["_perm"]
    d = o["_dist_sorted"].detach()
    uniq = torch.ones_like(d, dtype=torch.bool)
    uniq[:, 1:] &= d[:, 1:] != d[:, :-1]
    uniq[:, :-1] &= d[:, 1:] != d[:, :-1]
    assert torch.equal(perm.cpu().long()[uniq], p_ref[uniq])
    assert torch.equal(torch.sort(perm.cpu().long(), dim=1).values, torch.arange(N).expand(R, N))


# ------------------------------------------------------------------------------------------------ compositing
def _composite_inputs(case):
    o = case["out"]
    return o["_mlp_out"].detach(), o["_dist_sorted"].detach(), o["depth_volumes"].detach()


def test_composite_forward(case):
    lib = _capi.load()
    o = case["out"]
    logits, dist, z = _composite_inputs(case)
    R, N = dist.shape
    f = lambda *s: torch.empty(s, device=DEV)
    dens, al, w, dep, col, clo, wat = f(R, N), f(R, N), f(R, N), f(R), f(R, 3), f(R), f(R)
    ci = torch.empty((R,), dtype=torch.int32, device=DEV)
    _capi.check(lib.scenerf_hip_composite_forward(dv(logits).data_ptr(), dv(dist).data_ptr(), dv(z).data_ptr(), R, N, dens.data_ptr(),
                                                  al.data_ptr(), w.data_ptr(), dep.data_ptr(), col.data_ptr(), clo.data_ptr(),
                                                  wat.data_ptr(), ci.data_ptr(), _st()), "composite_forward")
    tol = dict(rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(dens.cpu(), o["densities"].detach(), **tol)
    torch.testing.assert_close(al.cpu(), o["alphas"].detach(), **tol)
    torch.testing.assert_close(w.cpu(), o["weights"].detach(), **tol)
    torch.testing.assert_close(dep.cpu(), o["depth"].detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(col.cpu(), o["color"].detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(clo.cpu(), o["closest_pts_to_depths"].detach(), rtol=1e-4, atol=2e-5)
    assert torch.equal(ci.cpu().long(), o["_closest_idx"]), "closest-sample index must be bit-exact"
    torch.testing.assert_close(wat.cpu(), o["weights_at_depth"].detach(), **tol)


@pytest.mark.parametrize("N", [64, 96, 128, 512])
def test_composite_forward_backward_vs_autograd(N):
    """All lane layouts (1/2/4/8 samples per lane, ragged N=96) against torch autograd of the oracle."""
    lib = _capi.load()
    gen = torch.Generator().manual_seed(N)
    R = 37
    logits = torch.randn(R * N, 4, generator=gen)
    logits[:, 3] -= 2.0
    dist = torch.sort(torch.rand(R, N, generator=gen) * 100 + 0.1, dim=1).values
    dist[:, 5] = dist[:, 4]  # a tie: delta = 0
    z = dist * 0.97
    lg = logits.clone().requires_grad_(True)
    dd = dist.clone().requires_grad_(True)
    zz = z.clone().requires_grad_(True)
    col = torch.sigmoid(lg[:, :3]).reshape(R, N, 3)
    den = orc.density_activation(lg[:, 3:4]).reshape(R, N)
    comp = orc.composite(den, dd.clone(), zz, col)
    gd, gc = torch.randn(R, generator=gen), torch.randn(R, 3, generator=gen)
    gw, ga = torch.randn(R, N, generator=gen) * 0.1, torch.randn(R, N, generator=gen) * 0.1
    gden, gz = torch.randn(R, N, generator=gen) * 0.1, torch.randn(R, N, generator=gen) * 0.1
    loss = (comp["depth"] * gd).sum() + (comp["color"] * gc).sum() + (comp["weights"] * gw).sum() + (comp["alphas"] * ga).sum() \
        + (den * gden).sum() + (zz * gz).sum()
    loss.backward()
    f = lambda *s: torch.empty(s, device=DEV)
    dens, al, w, dep, colr, clo, wat = f(R, N), f(R, N), f(R, N), f(R), f(R, 3), f(R), f(R)
    ci = torch.empty((R,), dtype=torch.int32, device=DEV)
    L, D, Z = dv(logits), dv(dist), dv(z)
    _capi.check(lib.scenerf_hip_composite_forward(L.data_ptr(), D.data_ptr(), Z.data_ptr(), R, N, dens.data_ptr(), al.data_ptr(),
                                                  w.data_ptr(), dep.data_ptr(), colr.data_ptr(), clo.data_ptr(), wat.data_ptr(),
                                                  ci.data_ptr(), _st()), "composite_forward")
    torch.testing.assert_close(w.cpu(), comp["weights"].detach(), rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(dep.cpu(), comp["depth"].detach(), rtol=2e-5, atol=2e-5)
    assert torch.equal(ci.cpu().long(), comp["closest_idx"])
    dl, ddist, dz = f(R * N, 4), f(R, N), f(R, N)
    _capi.check(lib.scenerf_hip_composite_backward(L.data_ptr(), D.data_ptr(), Z.data_ptr(), R, N, dv(gd).data_ptr(), dv(gc).data_ptr(),
                                                   dv(gw).data_ptr(), dv(ga).data_ptr(), dv(gden).data_ptr(), dv(gz).data_ptr(),
                                                   dl.data_ptr(), ddist.data_ptr(), dz.data_ptr(), _st()), "composite_backward")
    torch.testing.assert_close(dl.cpu(), lg.grad, rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(ddist.cpu(), dd.grad, rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(dz.cpu(), zz.grad, rtol=2e-4, atol=2e-5)


# ------------------------------------------------------------------------------------------------ RaySOM
def test_raysom_forward_and_sampler_backward(case):
    lib = _capi.load()
    g, o, ocfg = case["g"], case["out"], case["ocfg"]
    _, rcfg = _cfgs(g)
    cc = rcfg.to_c()
    R, N, G, U, P = g.pixels.shape[0], rcfg.n_samples, rcfg.n_gaussians, rcfg.n_pts_uni, rcfg.n_pts_per_gaussian
    gm, gs = o["gaussian_means"].detach(), o["gaussian_stds"].detach()
    dist, al = o["_dist_sorted"].detach(), o["alphas"].detach()
    f = lambda *s: torch.empty(s, device=DEV)
    lk, sm, sv, ks = f(R), f(R, G), f(R, G), f(R, G, 3)
    bmu = torch.empty((R, N), dtype=torch.uint8, device=DEV)
    _capi.check(lib.scenerf_hip_raysom_forward(C.byref(cc), dv(gm).data_ptr(), dv(gs).data_ptr(), dv(dist).data_ptr(), dv(al).data_ptr(), R,
                                               lk.data_ptr(), sm.data_ptr(), sv.data_ptr(), ks.data_ptr(), bmu.data_ptr(), _st()), "raysom_forward")
    # the two discrete choices on their own (BMU indices bit-exact): equal to the oracle's wherever the oracle's choice
    # is not a tie at rounding level (argmax margin > 1e-5 relative; a thresholded quantity > 1e-4 from its threshold)
    info = {}
    orc.ray_som_kl(gm, gs, dist.clone(), al, ocfg.som_sigma, ocfg.kl_std_floor, info=info)
    clear = info["bmu_margin"] > 1e-5
    assert bool((bmu.cpu().long()[clear] == o["_bmu"][clear]).all()) and float(clear.float().mean()) > 0.5
    clear_m = info["mask_margin"] > 1e-4
    assert bool((ks[:, :, 2].cpu()[clear_m] == info["mask"].float()[clear_m]).all())
    torch.testing.assert_close(sm.cpu(), o["som_means"].detach(), rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(sv.cpu(), o["som_vars"].detach(), rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(lk.cpu(), o["loss_kl"].detach(), rtol=2e-4, atol=2e-5)

    # sampler + KL backward vs autograd of the oracle's sampler/KL sub-graph
    offs = o["_offsets"].detach().clone().requires_grad_(True)
    anchors = orc.gaussian_anchor_distances(ocfg)
    means = torch.relu(anchors.reshape(1, G) + offs[:, :, 0]) + ocfg.gauss_floor
    stds = torch.relu(offs[:, :, 1] + ocfg.std) + ocfg.gauss_floor
    dg = means.repeat_interleave(P, dim=1) + g.noise_g * stds.repeat_interleave(P, dim=1)
    dg = torch.where(dg < 0.1, torch.full_like(dg, 0.1), dg)
    dall = torch.cat([o["_dist_u"], dg], dim=1)
    perm = o["_perm"]
    dsorted = torch.gather(dall, 1, perm)
    zsorted = dsorted * o["_unit"][:, 2:3]
    gen = torch.Generator().manual_seed(5)
    g_dd, g_dz = torch.randn(R, N, generator=gen), torch.randn(R, N, generator=gen)
    g_kl, g_gm, g_gs = torch.randn(R, generator=gen), torch.randn(R, G, generator=gen), torch.randn(R, G, generator=gen)
    kl, _, _, _ = orc.ray_som_kl(means, stds, dsorted, al, ocfg.som_sigma, ocfg.kl_std_floor)
    loss = (dsorted * g_dd).sum() + (zsorted * g_dz).sum() + (kl * g_kl).sum() + (means * g_gm).sum() + (stds * g_gs).sum()
    loss.backward()
    doff = f(R, G, 2)
    _capi.check(lib.scenerf_hip_sampler_backward(C.byref(cc), dv(o["_offsets"]).data_ptr(), dv(anchors).data_ptr(), dv(g.noise_g).data_ptr(),
                                                 dv(o["_unit"]).data_ptr(), dv(gm).data_ptr(), dv(gs).data_ptr(),
                                                 dv(perm.to(torch.int32)).data_ptr(), dv(g_dd).data_ptr(), dv(g_dz).data_ptr(),
                                                 ks.data_ptr(), dv(g_kl).data_ptr(), dv(g_gm).data_ptr(), dv(g_gs).data_ptr(), R,
                                                 doff.data_ptr(), _st()), "sampler_backward")
    torch.testing.assert_close(doff.cpu(), offs.grad, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("M", [4133, 40000])
@pytest.mark.parametrize("path", ["layers", "fused"])
def test_parameter_gradients_are_reproducible_run_to_run(M, path):
    """Every parameter gradient of scenerf_hip_mlp_backward on identical inputs, four runs: anything beyond fp32 atomic-ordering noise
    between two runs of the SAME path is a race.  (Regression test for the bias-gradient column sums of gemm_tn_kernel, which once came
    back wrong for 16 columns of one workgroup in a few hundred at M = 40,000: they were re-read from a staged LDS tile; they are now
    taken from the registers on their way into LDS.  M = 4,133: per-layer weight gradients; M = 40,000: the batched transposing-read
    launch, lin_in's padded tile included.)"""
    import dataclasses
    from scenerf_amd.renderer import MLP_PARAM_NAMES, PackedMLP, _MlpRun
    lib = _capi.load()
    rcfg = RenderConfig.kitti(precision="bf16", sphere_W=376, sphere_H=114)
    state = synth.mlp_state(102, 4)
    pk = PackedMLP([state[n].to(DEV) for n in MLP_PARAM_NAMES], 4, rcfg)
    gen = torch.Generator().manual_seed(M + 1)
    run = _MlpRun(M, 4, 1, DEV)
    run.Z.copy_((torch.randn(run.Z.shape, generator=gen) * 0.5).to(torch.bfloat16).to(DEV))
    xe = torch.zeros((M, 48))
    xe[:, :42] = torch.randn(M, 42, generator=gen).clamp(-1, 1)
    run.xenc.copy_(xe.to(DEV))
    run.tile_mask.fill_(31)
    cc = rcfg.to_c()
    _capi.check(lib.scenerf_hip_mlp_forward(C.byref(cc), C.byref(pk.c), run.Z.data_ptr(), run.xenc.data_ptr(), run.tile_mask.data_ptr(), M,
                                            C.byref(run.c), _st()), "fwd")
    dl = torch.randn(M, 4, generator=gen).to(DEV)
    tex = torch.full((M, 5, 4), -1, dtype=torch.int32, device=DEV)
    tw = torch.zeros((M, 5, 4), device=DEV)
    cc = dataclasses.replace(rcfg, fused_backward=(path == "fused")).to_c()
    outs = []
    for rep in range(4):
        gs = pk.grad_sink()
        pk.gflat.zero_()
        dH = torch.zeros((M, 2048), dtype=torch.bfloat16, device=DEV)
        dN = torch.zeros((3, M, 512), dtype=torch.bfloat16, device=DEV)
        _capi.check(lib.scenerf_hip_mlp_backward(C.byref(cc), C.byref(pk.c), C.byref(gs), run.Z.data_ptr(), run.xenc.data_ptr(),
                                                 run.tile_mask.data_ptr(), tex.data_ptr(), tw.data_ptr(), M, C.byref(run.c), dl.data_ptr(),
                                                 dH.data_ptr(), dN.data_ptr(), None, _st()), "bwd")
        torch.cuda.synchronize()
        outs.append([g.clone() for g in pk.unpack_grads()])
    worst = {}
    for rep in range(1, 4):
        for n, a, b in zip(MLP_PARAM_NAMES, outs[0], outs[rep]):
            worst[n] = max(worst.get(n, 0.0), float((a - b).norm() / max(float(a.norm()), 1e-20)))
    bad = {k: "%.1e" % v for k, v in worst.items() if v > 1e-5}
    assert not bad, "run-to-run differences beyond summation-order noise: %s" % bad
    # ... and the bias gradient of lin_in is the column sum of dH0 (fp64 on the GPU) of the last run
    ref = dH[:, :512].double().sum(0)
    got = outs[-1][MLP_PARAM_NAMES.index("lin_in.bias")].double()
    assert float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


@pytest.mark.parametrize("N,U,P", [(64, 32, 8), (96, 64, 8), (128, 64, 16), (512, 256, 64), (2, 2, 1)])
def test_fused_ray_tail_equals_the_stage_kernels(N, U, P):
    """scenerf_hip_ray_tail_forward / _backward (what RenderChunk launches: compositing + RaySOM in one kernel, their autograd + the
    sampler's in one kernel) against the four stage entries on the same inputs: every output bit-identical."""
    lib = _capi.load()
    G, R = 4, 301
    uniform_only = (N == U)
    rcfg = RenderConfig.bundlefusion(n_pts_uni=0, n_pts_per_gaussian=1) if uniform_only else RenderConfig.kitti(n_pts_uni=U, n_pts_per_gaussian=P)
    cc = rcfg.to_c()
    assert cc.n_samples == N
    gen = torch.Generator().manual_seed(N)
    logits = torch.randn(R * N, 4, generator=gen)
    logits[:, 3] -= 1.5
    dist = torch.sort(torch.rand(R, N, generator=gen) * 90 + 0.1, dim=1).values
    dist[:, N // 2] = dist[:, N // 2 - 1] if N > 2 else dist[:, N // 2]       # a tie
    z = dist * 0.97
    gm = torch.sort(torch.rand(R, G, generator=gen) * 80 + 2, dim=1).values
    gs = torch.rand(R, G, generator=gen) * 4 + 1.5
    perm = torch.stack([torch.randperm(N, generator=gen) for _ in range(R)]).to(torch.int32)
    offs = torch.randn(R, G, 2, generator=gen)
    anchors = torch.linspace(12.5, 87.5, G)
    noise = torch.randn(R, G * max(P, 1), generator=gen)
    unit = torch.nn.functional.normalize(torch.randn(R, 3, generator=gen), dim=1)
    ups = [torch.randn(s, generator=gen) for s in ((R,), (R, 3), (R, N), (R, N), (R, N), (R, N), (R,), (R, G), (R, G))]
    gd, gc, gw, ga, gden, gz, gkl, ggm, ggs = [dv(t) for t in ups]
    L, D, Z, GM, GS, PM, OF, AN, NZ, UN = [dv(t) for t in (logits, dist, z, gm, gs, perm, offs, anchors, noise, unit)]
    f = lambda *s_: torch.empty(s_, device=DEV)

    def outs():
        return dict(dens=f(R, N), al=f(R, N), w=f(R, N), dep=f(R), col=f(R, 3), clo=f(R), wat=f(R), ci=torch.empty(R, dtype=torch.int32, device=DEV),
                    lk=f(R), sm=f(R, G), sv=f(R, G), ks=f(R, G, 3), bmu=torch.empty((R, N), dtype=torch.uint8, device=DEV))
    a, b = outs(), outs()
    _capi.check(lib.scenerf_hip_composite_forward(L.data_ptr(), D.data_ptr(), Z.data_ptr(), R, N, a["dens"].data_ptr(), a["al"].data_ptr(),
                                                  a["w"].data_ptr(), a["dep"].data_ptr(), a["col"].data_ptr(), a["clo"].data_ptr(),
                                                  a["wat"].data_ptr(), a["ci"].data_ptr(), _st()), "composite_forward")
    _capi.check(lib.scenerf_hip_raysom_forward(C.byref(cc), GM.data_ptr(), GS.data_ptr(), D.data_ptr(), a["al"].data_ptr(), R, a["lk"].data_ptr(),
                                               a["sm"].data_ptr(), a["sv"].data_ptr(), a["ks"].data_ptr(), a["bmu"].data_ptr(), _st()), "raysom_forward")
    _capi.check(lib.scenerf_hip_ray_tail_forward(C.byref(cc), L.data_ptr(), D.data_ptr(), Z.data_ptr(), GM.data_ptr(), GS.data_ptr(), R,
                                                 b["dens"].data_ptr(), b["al"].data_ptr(), b["w"].data_ptr(), b["dep"].data_ptr(), b["col"].data_ptr(),
                                                 b["clo"].data_ptr(), b["wat"].data_ptr(), b["ci"].data_ptr(), b["lk"].data_ptr(), b["sm"].data_ptr(),
                                                 b["sv"].data_ptr(), b["ks"].data_ptr(), b["bmu"].data_ptr(), _st()), "ray_tail_forward")
    for k_ in a:
        assert torch.equal(a[k_], b[k_]), "forward output %s differs" % k_
    # ... and the compositing-only instantiation (loss_kl == NULL: what a depth / colour render under no_grad launches): the same values,
    # the (R, N) outputs written only where asked for
    c = outs()
    c["w"].fill_(-7.0)
    _capi.check(lib.scenerf_hip_ray_tail_forward(C.byref(cc), L.data_ptr(), D.data_ptr(), Z.data_ptr(), None, None, R,
                                                 None, c["al"].data_ptr(), None, c["dep"].data_ptr(), c["col"].data_ptr(),
                                                 c["clo"].data_ptr(), c["wat"].data_ptr(), c["ci"].data_ptr(), None, None, None, None, None, _st()),
                "ray_tail_forward (no SOM)")
    for k_ in ("al", "dep", "col", "clo", "wat", "ci"):
        assert torch.equal(a[k_], c[k_]), "compositing-only output %s differs" % k_
    assert float(c["w"].min()) == -7.0 and float(c["w"].max()) == -7.0
    assert lib.scenerf_hip_ray_tail_forward(C.byref(cc), L.data_ptr(), D.data_ptr(), Z.data_ptr(), GM.data_ptr(), GS.data_ptr(), R, None, None, None,
                                            c["dep"].data_ptr(), c["col"].data_ptr(), c["clo"].data_ptr(), c["wat"].data_ptr(), c["ci"].data_ptr(),
                                            None, c["sm"].data_ptr(), None, None, None, _st()) != 0       # half of the RaySOM outputs: refused
    dl1, dd1, dz1, do1 = f(R * N, 4), f(R, N), f(R, N), f(R, G, 2)
    dl2, dd2, dz2, do2, do3 = f(R * N, 4), f(R, N), f(R, N), f(R, G, 2), f(R, G, 2)
    _capi.check(lib.scenerf_hip_composite_backward(L.data_ptr(), D.data_ptr(), Z.data_ptr(), R, N, gd.data_ptr(), gc.data_ptr(), gw.data_ptr(),
                                                   ga.data_ptr(), gden.data_ptr(), gz.data_ptr(), dl1.data_ptr(), dd1.data_ptr(), dz1.data_ptr(), _st()),
                "composite_backward")
    _capi.check(lib.scenerf_hip_sampler_backward(C.byref(cc), OF.data_ptr(), AN.data_ptr(), NZ.data_ptr(), UN.data_ptr(), GM.data_ptr(), GS.data_ptr(),
                                                 PM.data_ptr(), dd1.data_ptr(), dz1.data_ptr(), a["ks"].data_ptr(), gkl.data_ptr(), ggm.data_ptr(),
                                                 ggs.data_ptr(), R, do1.data_ptr(), _st()), "sampler_backward")
    for (o_, dd_, dz_) in ((do2, dd2, dz2), (do3, None, None)):
        _capi.check(lib.scenerf_hip_ray_tail_backward(C.byref(cc), L.data_ptr(), D.data_ptr(), Z.data_ptr(), R, gd.data_ptr(), gc.data_ptr(),
                                                      gw.data_ptr(), ga.data_ptr(), gden.data_ptr(), gz.data_ptr(), OF.data_ptr(), AN.data_ptr(),
                                                      NZ.data_ptr(), UN.data_ptr(), GM.data_ptr(), GS.data_ptr(), PM.data_ptr(), a["ks"].data_ptr(),
                                                      gkl.data_ptr(), ggm.data_ptr(), ggs.data_ptr(), dl2.data_ptr(), o_.data_ptr(),
                                                      _capi.ptr(dd_), _capi.ptr(dz_), None, None, None, _st()), "ray_tail_backward")
    assert torch.equal(dl1, dl2) and torch.equal(dd1, dd2) and torch.equal(dz1, dz2)
    assert torch.equal(do1, do2) and torch.equal(do1, do3)
    assert bool(torch.isfinite(do1).all()) and float(do1.abs().max()) > 0


# ------------------------------------------------------------------------------------------------ MLP pass
def _mlp_case(case, precision, which):
    """Feed the oracle's x_in to the HIP MLP pass; returns everything needed to compare."""
    from scenerf_amd.renderer import MLP_PARAM_NAMES, PackedMLP
    o = case["out"]
    g = case["g"]
    _, rcfg = _cfgs(g, precision)
    xin = (o["_xin"] if which == "mlp" else o["_xin_g"]).detach()
    state = case["mlp"] if which == "mlp" else case["mlpg"]
    d_out = 4 if which == "mlp" else 2
    params = [dv(state[n]) for n in MLP_PARAM_NAMES]
    pk = PackedMLP(params, d_out, rcfg)
    return rcfg, xin, state, d_out, pk


@pytest.mark.parametrize("precision,which", [("fp32", "mlp"), ("fp32", "gauss"), ("bf16", "mlp")])
def test_mlp_forward_backward(case, precision, which):
    from scenerf_amd.renderer import MLP_PARAM_NAMES, _MlpRun
    lib = _capi.load()
    rcfg, xin, state, d_out, pk = _mlp_case(case, precision, which)
    cc = rcfg.to_c()
    prec = rcfg.precision_code
    act = torch.bfloat16 if prec else torch.float32
    M = xin.shape[0]
    run = _MlpRun(M, d_out, prec, torch.device(DEV))
    run.Z.zero_()
    run.Z[:M] = dv(xin[:, :2480], act)
    xe = torch.zeros((M, 48))
    xe[:, :42] = xin[:, 2480:]
    run.xenc.copy_(dv(xe))
    run.tile_mask.fill_(31)
    _capi.check(lib.scenerf_hip_mlp_forward(C.byref(cc), C.byref(pk.c), run.Z.data_ptr(), run.xenc.data_ptr(), run.tile_mask.data_ptr(), M,
                                            C.byref(run.c), _st()), "mlp_forward")
    # oracle with autograd
    p = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    xr = xin.clone().requires_grad_(True)
    keep = {}
    ref = orc.resnetfc_forward(p, xr, keep=keep)
    scale = float(ref.abs().max())
    tol = 2e-5 if prec == 0 else 3e-2
    err = float((run.logits.cpu() - ref.detach()).abs().max())
    print("%s %s logits max err %.3e (scale %.2f)" % (precision, which, err, scale))
    assert err <= tol * max(scale, 1.0)
    for b in range(4):
        e = float((run.H[b].float().cpu() - keep["h%d" % b].detach()).abs().max())
        assert e <= tol * max(float(keep["h%d" % b].abs().max()), 1.0), "H%d err %.3e" % (b, e)
    # backward: random upstream, no map scatter here (features' gradient checked end-to-end)
    gen = torch.Generator().manual_seed(3)
    dl = torch.randn(M, d_out, generator=gen)
    (ref * dl).sum().backward()
    gsink = pk.grad_sink()
    dH = torch.empty((M, 2048), dtype=act, device=DEV)
    dN = torch.empty((3, M, 512), dtype=act, device=DEV)
    tex = torch.full((M, 5, 4), -1, dtype=torch.int32, device=DEV)
    tw = torch.zeros((M, 5, 4), device=DEV)
    _capi.check(lib.scenerf_hip_mlp_backward(C.byref(cc), C.byref(pk.c), C.byref(gsink), run.Z.data_ptr(), run.xenc.data_ptr(),
                                             run.tile_mask.data_ptr(), tex.data_ptr(), tw.data_ptr(), M, C.byref(run.c),
                                             dv(dl).data_ptr(), dH.data_ptr(), dN.data_ptr(), None, _st()), "mlp_backward")
    grads = dict(zip(MLP_PARAM_NAMES, pk.unpack_grads()))
    # gradients: relative L2 error per parameter (bf16 operands: activations AND upstream gradients are rounded to
    # 8 bits at every layer, so the deepest gradients carry a few percent), plus a loose element-wise bound
    gtol = 2e-4 if prec == 0 else 8e-2
    worst = {}
    for n in MLP_PARAM_NAMES:
        r = p[n].grad
        got = grads[n].cpu()
        rel = float((got - r).norm() / max(float(r.norm()), 1e-12))
        e = float((got - r).abs().max())
        s = float(r.abs().max())
        worst[n] = (rel, e / max(s, 1e-12))
        assert rel <= gtol, "%s: relative L2 grad error %.3e" % (n, rel)
        assert e <= 4 * gtol * max(s, 1e-6), "%s: grad err %.3e vs scale %.3e" % (n, e, s)
    print("worst grad errors:", sorted(worst.items(), key=lambda kv: -kv[1][0])[:3])


@pytest.mark.parametrize("M", [64, 5000, 40000])
def test_mlp_forward_fused_matches_layer_path(case, M):
    """fused.hip (one kernel for the whole trunk, residual stream in fp32 registers) against the per-layer GEMM path on the
    same packed weights: saved activations are relu(H_b) / relu(N_b) -- what the backward pass consumes -- and must agree
    with the layer path's within bf16 rounding of the residual stream; logits against the fp32 oracle must be at least
    as close as the layer path's.  Tiles use different scale masks (skipped K segments)."""
    import dataclasses
    from scenerf_amd.renderer import _MlpRun
    lib = _capi.load()
    rcfg, xin0, state, d_out, pk = _mlp_case(case, "bf16", "mlp")
    gen = torch.Generator().manual_seed(M)
    z = torch.randn(M, 2480, generator=gen) * 0.5
    xe = torch.zeros((M, 48))
    xe[:, :42] = torch.randn(M, 42, generator=gen).clamp(-1, 1)
    ntile = (M + 127) // 128
    masks = torch.tensor([7, 31, 1, 5, 0, 24, 3], dtype=torch.uint8)[torch.arange(ntile) % 7]
    seg = [0]
    for c, _, _ in rcfg.map_shapes():
        seg.append(seg[-1] + c)
    for t in range(ntile):
        for s_ in range(5):
            if not (int(masks[t]) >> s_) & 1:
                z[t * 128:(t + 1) * 128, seg[s_]:seg[s_ + 1]] = 0
    runs = {}
    for name, min_rows in (("layers", -1), ("fused", 1)):
        cc = dataclasses.replace(rcfg, fused_min_rows=min_rows).to_c()   # kernel path = explicit call state (scenerf_cfg.fused_min_rows)
        run = _MlpRun(M, d_out, 1, torch.device(DEV))
        run.Z.zero_()
        run.Z[:M] = dv(z, torch.bfloat16)
        run.xenc.copy_(dv(xe))
        run.tile_mask.zero_()
        run.tile_mask[:ntile] = dv(masks)
        _capi.check(lib.scenerf_hip_mlp_forward(C.byref(cc), C.byref(pk.c), run.Z.data_ptr(), run.xenc.data_ptr(), run.tile_mask.data_ptr(),
                                                M, C.byref(run.c), _st()), "mlp_forward")
        torch.cuda.synchronize()
        runs[name] = run
    a, b = runs["layers"], runs["fused"]
    xin = torch.cat([run.Z[:M].float().cpu(), xe[:, :42]], dim=1)
    keep = {}
    ref = orc.resnetfc_forward(state, xin, keep=keep)
    for i in range(4):
        want = torch.relu(a.H[i].float())
        got = b.H[i].float()
        assert float((got < 0).sum()) == 0
        scale = float(want.abs().max())
        e = float((got - want).abs().max())
        e_or = float((got.cpu() - torch.relu(keep["h%d" % i])).abs().max())
        e_or_layers = float((want.cpu() - torch.relu(keep["h%d" % i])).abs().max())
        print("H%d: fused vs layers %.3e, fused vs oracle %.3e, layers vs oracle %.3e (scale %.2f)" % (i, e, e_or, e_or_layers, scale))
        assert e <= 3e-2 * max(scale, 1.0)
        assert e_or <= max(1.25 * e_or_layers, 1e-2 * max(scale, 1.0))
    for i in range(3):
        want = torch.relu(a.Nn[i].float())
        got = b.Nn[i].float()
        e = float((got - want).abs().max())
        assert e <= 3e-2 * max(float(want.abs().max()), 1.0), "N%d err %.3e" % (i, e)
    el = float((a.logits.cpu() - ref).abs().max())
    ef = float((b.logits.cpu() - ref).abs().max())
    print("logits err: layers %.3e fused %.3e" % (el, ef))
    assert ef <= max(1.25 * el, 1e-2 * max(float(ref.abs().max()), 1.0))


@pytest.mark.parametrize("M,lean", [(64, False), (4133, False), (40000, False), (40000, True)])
def test_mlp_forward_wide_kernel_matches_ring_kernel(case, M, lean):
    """wide.hip (128-row blocks, one wave per SIMD, accumulators in the accumulator file) against fused.hip's ring kernel: the same
    rounding points (H_b rounded to bf16 once per block), but the bias is added after the K sum instead of before it (and layer 0
    sums its K segments in a different order), so a saved activation may differ in its last bf16 bit -- and such a difference propagates through the following layers
    like any bf16 rounding: >= 99 % of the elements bit-identical, every element within 2 % of the tensor's scale (the fused-vs-layers
    test allows 3 %), sign bits >= 99.9 % identical, logits within 1 % of their scale, and against the fp32 oracle the kernel is as
    close as the ring kernel (within 25 %).  Mixed tile masks, a ragged tail, the lean inference buffers."""
    import dataclasses
    from scenerf_amd.renderer import _MlpRun
    lib = _capi.load()
    rcfg, xin0, state, d_out, pk = _mlp_case(case, "bf16", "mlp")
    gen = torch.Generator().manual_seed(M + 7)
    ntile = (M + 127) // 128
    masks = torch.tensor([7, 31, 1, 5, 0, 24, 3, 16], dtype=torch.uint8)[torch.arange(ntile) % 8]
    nz = min(M, 40000)
    z = torch.randn(nz, 2480, generator=gen).to(torch.bfloat16)
    xe = torch.randn(nz, 48, generator=gen).clamp(-1, 1)
    xe[:, 42:] = 0
    seg = [0]
    for c, _, _ in rcfg.map_shapes():
        seg.append(seg[-1] + c)
    reps = (M + nz - 1) // nz
    runs = {}
    for name in ("ring", "wide"):
        cc = dataclasses.replace(rcfg, fused_min_rows=1, fwd_kernel=name, wide_any_m=True).to_c()   # (SCENERF_FLAG_WIDE_ANY_M: also at small M)
        run = _MlpRun(M, d_out, 1, torch.device(DEV), lean=lean)
        run.Z.fill_(float("nan"))      # columns of scales a tile does not touch must never be read (beyond the dense first 256)
        run.Z[:, :256] = 0
        zz = dv(z).repeat(reps, 1)[:M]
        for s_ in range(5):
            act = ((masks.long() >> s_) & 1).bool().repeat_interleave(128)[:M]
            cols = slice(seg[s_], seg[s_ + 1])
            run.Z[:M, cols] = torch.where(dv(act)[:, None], zz[:, cols], run.Z[:M, cols])
        run.xenc.copy_(dv(xe).repeat(reps, 1)[:M])
        run.tile_mask.zero_()
        run.tile_mask[:ntile] = dv(masks)
        _capi.check(lib.scenerf_hip_mlp_forward(C.byref(cc), C.byref(pk.c), run.Z.data_ptr(), run.xenc.data_ptr(), run.tile_mask.data_ptr(),
                                                M, C.byref(run.c), _st()), "mlp_forward")
        torch.cuda.synchronize()
        runs[name] = run
    a, b = runs["ring"], runs["wide"]
    assert torch.isfinite(b.logits).all()
    scale = float(a.logits.abs().max())
    assert float((a.logits - b.logits).abs().max()) <= 1e-2 * max(scale, 1.0)
    if not lean:
        for nm, x, y in [("H%d" % i, a.H[i], b.H[i]) for i in range(4)] + [("N%d" % i, a.Nn[i], b.Nn[i]) for i in range(3)]:
            x, y = x.float(), y.float()
            assert float((x - y).abs().max()) <= 2e-2 * float(x.abs().max()), (nm, float((x - y).abs().max()), float(x.abs().max()))
            assert float((x == y).float().mean()) >= 0.99, (nm, float((x == y).float().mean()))
        same = (a.sign_bits[:6, :M] == b.sign_bits[:6, :M]).float().mean()
        assert float(same) >= 0.999, float(same)
    # against the fp32 oracle on the same (bf16-rounded) inputs: no further from it than the ring kernel
    mo = min(M, 4096)
    zin = torch.where(torch.isnan(a.Z[:mo].float()), torch.zeros((), device=DEV), a.Z[:mo].float()).cpu()
    for s_ in range(5):
        act = ((masks.long() >> s_) & 1).bool().repeat_interleave(128)[:mo]
        zin[~act, seg[s_]:seg[s_ + 1]] = 0
    ref = orc.resnetfc_forward(state, torch.cat([zin, a.xenc[:mo, :42].cpu()], dim=1))
    ea, eb = float((a.logits[:mo].cpu() - ref).abs().max()), float((b.logits[:mo].cpu() - ref).abs().max())
    print("logits vs oracle: ring %.3e wide %.3e" % (ea, eb))
    assert eb <= max(1.25 * ea, 1e-2 * max(float(ref.abs().max()), 1.0))


@pytest.mark.parametrize("M", [4096 + 37, 40000])
def test_mlp_backward_fused_matches_layer_path(case, M):
    """fused.hip MODE 1 (the six dgrad GEMMs of the residual blocks in one kernel, sign gates rebuilt from the saved activations by
    the forward-emitted sign bits) against the per-layer dgrad GEMMs on the same forward state: dH column blocks 0..2 and dN must be
    bit-identical, every parameter gradient equal up to the summation order of the fp32 atomics."""
    import dataclasses
    from scenerf_amd.renderer import MLP_PARAM_NAMES, _MlpRun
    lib = _capi.load()
    rcfg, xin0, state, d_out, pk = _mlp_case(case, "bf16", "mlp")
    cc = rcfg.to_c()
    gen = torch.Generator().manual_seed(M + 1)
    run = _MlpRun(M, d_out, 1, torch.device(DEV))
    run.Z.copy_(dv(torch.randn(run.Z.shape, generator=gen) * 0.5, torch.bfloat16))
    xe = torch.zeros((M, 48))
    xe[:, :42] = torch.randn(M, 42, generator=gen).clamp(-1, 1)
    run.xenc.copy_(dv(xe))
    run.tile_mask.fill_(31)
    _capi.check(lib.scenerf_hip_mlp_forward(C.byref(cc), C.byref(pk.c), run.Z.data_ptr(), run.xenc.data_ptr(), run.tile_mask.data_ptr(), M,
                                            C.byref(run.c), _st()), "mlp_forward")
    dl = dv(torch.randn(M, d_out, generator=gen))
    tex = torch.full((M, 5, 4), -1, dtype=torch.int32, device=DEV)
    tw = torch.zeros((M, 5, 4), device=DEV)
    res = {}
    for name in ("layers", "fused", "wide_staged", "wide"):
        # scenerf_cfg.flags: SCENERF_FLAG_NO_FUSED_BWD (per-layer dgrad GEMMs) / SCENERF_FLAG_WIDE_BWD (wide.hip's 128-row chain, which makes
        # lin_out's input gradient itself) / SCENERF_FLAG_WIDE_BWD_STAGED (... on linout_bwd's dH3, like the ring kernel)
        cc = dataclasses.replace(rcfg, fused_backward=(name != "layers"), bwd_kernel=name if name.startswith("wide") else "ring", wide_any_m=True).to_c()
        gs = pk.grad_sink()
        pk.gflat.zero_()
        dH = torch.zeros((M, 2048), dtype=torch.bfloat16, device=DEV)
        dN = torch.zeros((3, M, 512), dtype=torch.bfloat16, device=DEV)
        _capi.check(lib.scenerf_hip_mlp_backward(C.byref(cc), C.byref(pk.c), C.byref(gs), run.Z.data_ptr(), run.xenc.data_ptr(),
                                                 run.tile_mask.data_ptr(), tex.data_ptr(), tw.data_ptr(), M, C.byref(run.c),
                                                 dl.data_ptr(), dH.data_ptr(), dN.data_ptr(), None, _st()), "mlp_backward")
        torch.cuda.synchronize()
        res[name] = (dH.float().cpu(), dN.float().cpu(), [g.clone().cpu() for g in pk.unpack_grads()])
    dHa, dNa, ga = res["layers"]
    for other in ("fused", "wide_staged"):
        dHb, dNb, gb = res[other]
        assert torch.equal(dHa[:, 1536:], dHb[:, 1536:])          # lin_out's backward is shared
        bad = []
        for b in (2, 1, 0):   # chain order
            for nm, x, y in (("dN%d" % b, dNa[b], dNb[b]), ("dH%d" % b, dHa[:, 512 * b:512 * (b + 1)], dHb[:, 512 * b:512 * (b + 1)])):
                # same K order, same rounding points (dH / dN to bf16 per block), same sign gates: both chains are BIT-IDENTICAL to the six
                # per-layer dgrad GEMMs
                if not torch.equal(x, y):
                    bad.append("%s %s: rel L2 %.3e" % (other, nm, float((x - y).norm() / max(float(x.norm()), 1e-20))))
        assert not bad, bad
        # the weight gradients are summed with fp32 atomics over M splits in both paths: equal up to summation order
        rels = {n: float((x - y).norm() / max(float(x.norm()), 1e-20)) for n, x, y in zip(MLP_PARAM_NAMES, ga, gb)}
        print("%s vs layers parameter gradients, rel L2:" % other, {k: "%.1e" % v for k, v in rels.items()})
        for n, rel in rels.items():
            assert rel <= 1e-4, "%s %s: relative L2 difference %.3e" % (other, n, rel)
    # the default 128-row chain makes dH3 = (d_logits W_out) * [H3 > 0] in its prologue, on the matrix cores (three-term bf16 splits of
    # both fp32 operands: every term product exact, fp32 accumulation), where linout_bwd runs an fp32 FMA chain: the two agree to fp32
    # rounding of the four-term sum, i.e. the bf16 results are EQUAL except where the exact value sits within that rounding of a bf16
    # rounding boundary (measured 1.6-1.8e-5 of the elements; such an element is one bf16 ulp off, more only where the four products
    # cancel and fp32 rounding of the TERMS exceeds an ulp of the small sum).  Against the exact product (float64 on the same operands)
    # neither is further away than half a bf16 ulp plus that fp32 rounding.  The rest of the chain is the same instruction stream as
    # wide_staged: its outputs follow dH3 (a flipped element moves them like any bf16 rounding does).
    dHw, dNw, gw = res["wide"]
    x, y = dHa[:, 1536:], dHw[:, 1536:]
    diff = x != y
    frac = float(diff.float().mean())
    print("prologue dH3 vs linout_bwd dH3: %.2e of the elements differ" % frac)
    assert frac <= 2e-4, frac
    gate = (run.H[3].float().cpu() > 0)
    exact = (dl.double().cpu() @ state["lin_out.weight"].double()) * gate
    terms = (dl.double().cpu().abs() @ state["lin_out.weight"].double().abs()) * gate
    if bool(diff.any()):
        ulp = torch.maximum(x.abs(), y.abs())[diff] * 2.0 ** -7 + terms[diff].float() * 1e-6     # one bf16 ulp of the larger one + fp32 rounding of the terms
        assert bool(((x - y).abs()[diff] <= ulp).all())
    for nm, got in (("linout_bwd", x), ("prologue", y)):
        err = (got.double() - exact).abs()
        bound = exact.abs() * 2.0 ** -8 + terms * 4e-7 + 1e-30    # half a bf16 ulp (8 significant bits) + fp32 rounding of the four-term sum
        assert bool((err <= bound).all()), (nm, float((err / bound.clamp(min=1e-30)).max()))
    for b in (2, 1, 0):
        for nm, x, y in (("dN%d" % b, dNa[b], dNw[b]), ("dH%d" % b, dHa[:, 512 * b:512 * (b + 1)], dHw[:, 512 * b:512 * (b + 1)])):
            rel = float((x - y).norm() / max(float(x.norm()), 1e-20))
            assert rel <= 2e-3 and float((x == y).float().mean()) >= 0.99, (nm, rel, float((x == y).float().mean()))
    rels = {n: float((x - y).norm() / max(float(x.norm()), 1e-20)) for n, x, y in zip(MLP_PARAM_NAMES, ga, gw)}
    print("wide (prologue) vs layers parameter gradients, rel L2:", {k: "%.1e" % v for k, v in rels.items()})
    for n, rel in rels.items():
        assert rel <= 1e-3, "wide %s: relative L2 difference %.3e" % (n, rel)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_direct_chw_scales_match_the_converted_path(case, precision):
    """scenerf_cfg.map_chw (RenderConfig.direct_scales): a pyramid level read from the caller's fp32 (C,H,W) tensor and scattered into a
    (C,H,W) gradient buffer must give what the (H,W,C) copy + accumulator + transpose give.  The sample indices are drawn small
    enough to be in range of EVERY scale (quirk Q1 keeps the coarse ones out of range in the real geometry, so nothing else
    exercises their taps)."""
    import dataclasses
    from scenerf_amd.renderer import MLP_PARAM_NAMES, PackedMLP
    lib = _capi.load()
    g = case["g"]
    _, rcfg0 = _cfgs(g, precision)
    prec = rcfg0.precision_code
    act = torch.bfloat16 if prec else torch.float32
    shapes = rcfg0.map_shapes()
    gen = torch.Generator().manual_seed(17)
    M = 1000
    W16, H16 = rcfg0.sphere_W // 16, rcfg0.sphere_H // 16
    idx = torch.stack([torch.randint(0, W16 + 2, (M,), generator=gen), torch.randint(0, H16 + 2, (M,), generator=gen)], dim=1).to(torch.int32)
    idx[::7] = torch.tensor([rcfg0.sphere_W - 1, rcfg0.sphere_H - 1], dtype=torch.int32)   # some rows out of range of the coarse scales
    maps = [dv(case["maps"][k]) for k in ["1_1", "1_2", "1_4", "1_8", "1_16"]]
    params = [dv(case["mlp"][n]) for n in MLP_PARAM_NAMES]
    dH = dv((torch.randn(M, 2048, generator=gen) * 0.1), act)
    Mpad = (M + 127) // 128 * 128
    res = {}
    for name, direct in (("converted", ()), ("direct", (2, 3, 4))):
        rcfg = dataclasses.replace(rcfg0, direct_scales=direct)
        cc = rcfg.to_c()
        srcs = []
        for i, ((c, h, w), src) in enumerate(zip(shapes, maps)):
            if i in direct:
                srcs.append(src)
            else:
                dst = torch.empty((h, w, c), dtype=act, device=DEV)
                _capi.check(lib.scenerf_hip_maps_chw_to_hwc(src.data_ptr(), dst.data_ptr(), c, h, w, prec, _st()), "maps_chw_to_hwc")
                srcs.append(dst)
        Z = torch.zeros((Mpad, 2480), dtype=act, device=DEV)
        mask = torch.zeros((Mpad // 128,), dtype=torch.uint8, device=DEV)
        tex = torch.empty((M, 5, 4), dtype=torch.int32, device=DEV)
        tw = torch.empty((M, 5, 4), device=DEV)
        arr = (C.c_void_p * 5)(*[t.data_ptr() for t in srcs])
        _capi.check(lib.scenerf_hip_gather_features(C.byref(cc), C.byref(arr), dv(idx).data_ptr(), M, Z.data_ptr(), mask.data_ptr(),
                                                    tex.data_ptr(), tw.data_ptr(), _st()), "gather_features")
        assert int(mask.max()) == 31, "the drawn indices must reach every scale"
        pk = PackedMLP(params, 4, rcfg)
        gm = [torch.zeros((c, h, w) if i in direct else (h, w, c), device=DEV) for i, (c, h, w) in enumerate(shapes)]
        garr = (C.c_void_p * 5)(*[t.data_ptr() for t in gm])
        _capi.check(lib.scenerf_hip_mlp_feature_grads(C.byref(cc), C.byref(pk.c), mask.data_ptr(), tex.data_ptr(), tw.data_ptr(), M,
                                                      dH.data_ptr(), C.byref(garr), _st()), "mlp_feature_grads")
        grads = []
        for i, (c, h, w) in enumerate(shapes):
            if i in direct:
                grads.append(gm[i])
            else:
                out = torch.empty((c, h, w), device=DEV)
                _capi.check(lib.scenerf_hip_grads_hwc_to_chw(gm[i].data_ptr(), out.data_ptr(), c, h, w, _st()), "grads_hwc_to_chw")
                grads.append(out)
        torch.cuda.synchronize()
        res[name] = (Z[:M].float().cpu(), [x.cpu() for x in grads])
    (Za, ga), (Zb, gb) = res["converted"], res["direct"]
    if prec == 0:
        assert torch.equal(Za, Zb)                      # same fp32 values, same blend order
    else:
        torch.testing.assert_close(Zb, Za, rtol=1.2e-2, atol=1.2e-2)   # the direct path blends un-rounded fp32 map values
    for i, (x, y) in enumerate(zip(ga, gb)):
        assert float(x.abs().max()) > 0, "scale %d received no gradient" % i
        rel = float((x - y).norm() / x.norm())
        assert rel <= (1e-5 if prec == 0 else 2e-2), "scale %d: relative L2 %.3e" % (i, rel)


@pytest.mark.parametrize("M", [193 * 128 - 41, 40000])
def test_feature_gradient_kernel_matches_gemm_scatter_epilogue(M):
    """dfeat.hip (one workgroup per 128-row tile, dH staged once, per-tap run sums before the atomics) against the GEMM family's scatter
    epilogue (scenerf_cfg.flags & SCENERF_FLAG_DFEAT_GEMM) on the same dH and taps: every level's (H,W,C) gradient map equal up to fp32
    summation order.  Ragged last tile, all 32 tile masks, taps that repeat along a ray / differ from row to row / are absent (-1),
    zero weights."""
    import dataclasses
    from scenerf_amd.renderer import MLP_PARAM_NAMES, PackedMLP
    lib = _capi.load()
    rcfg = RenderConfig.kitti(precision="bf16", sphere_W=376, sphere_H=114)
    state = synth.mlp_state(5, 4)
    pk = PackedMLP([torch.as_tensor(state[n]).to(DEV) for n in MLP_PARAM_NAMES], 4, rcfg)
    gen = torch.Generator().manual_seed(M)
    ntile = (M + 127) // 128
    masks = (torch.arange(ntile) % 32).to(torch.uint8)
    masks[::5] = 1                                                     # plenty of finest-level-only tiles, like the real geometry
    dH = dv((torch.randn(M, 2048, generator=gen) * 0.1), torch.bfloat16)
    shapes = rcfg.map_shapes()
    tex = torch.full((M, 5, 4), -1, dtype=torch.int32)
    tw = torch.zeros((M, 5, 4))
    r = torch.arange(M)
    for s_, (c, h, w) in enumerate(shapes):
        act = ((masks.long() >> s_) & 1).bool().repeat_interleave(128)[:M]
        walk = torch.randint(0, max(w - 12, 1), (ntile,), generator=gen).repeat_interleave(128)[:M] + (r % 128) * 9 // 128
        jump = torch.randint(0, w - 1, (M,), generator=gen)
        x0 = torch.where((r // 128) % 3 == 0, jump, walk).clamp(0, w - 2)   # every third tile: unrelated texels row by row
        y0 = torch.randint(0, max(h - 1, 1), (ntile,), generator=gen).repeat_interleave(128)[:M].clamp(0, h - 2)
        for k, (dx, dy) in enumerate(((0, 0), (1, 0), (0, 1), (1, 1))):
            t = ((y0 + dy) * w + x0 + dx).int()
            drop = (t.long() * 7919 + k) % 10 == 0                      # out-of-range taps (again a function of the position)
            tex[:, s_, k] = torch.where(act & ~drop, t, torch.full_like(t, -1))
            # (a row's weights are a function of its sample position, like the forward gather's: rows with the same taps have the same
            # weights -- the GEMM epilogue's run merging relies on that; ~6 % of the positions get weight 0)
            wv = ((t.long() * 2654435761 + k * 40503) % 1000).float() / 1000.0
            tw[:, s_, k] = torch.where(wv < 0.06, torch.zeros(M), wv)
    tex, tw, masks_d = dv(tex), dv(tw), dv(masks)
    res = {}
    for name in ("gemm", "dfeat"):
        cc = dataclasses.replace(rcfg, dfeat_gemm=(name == "gemm")).to_c()
        gm = [torch.zeros((h, w, c), device=DEV) for (c, h, w) in shapes]
        arr = (C.c_void_p * 5)(*[g.data_ptr() for g in gm])
        _capi.check(lib.scenerf_hip_mlp_feature_grads(C.byref(cc), C.byref(pk.c), masks_d.data_ptr(), tex.data_ptr(), tw.data_ptr(), M,
                                                      dH.data_ptr(), arr, _st()), "mlp_feature_grads")
        torch.cuda.synchronize()
        res[name] = gm
    for i, (a, b) in enumerate(zip(res["gemm"], res["dfeat"])):
        scale = float(a.abs().max())
        assert scale > 0, "level %d must be exercised" % i
        err = float((a - b).abs().max())
        print("level %d: max |diff| %.2e of %.2e" % (i, err, scale))
        assert err <= 4e-6 * scale, (i, err, scale)   # measured <= 6.4e-7 (fp32 summation order)


def test_pack_in_two_calls_writes_the_same_operands():
    """SCENERF_FLAG_PACK_FORWARD then _REST (PackedMLP(split=True): a forward's operands first, on a pack stream, launch deferred) leaves the
    same bytes in every operand buffer as the one-call pack, and a zeroed gradient sink; wait_ready() orders a forward behind the first
    call, wait_ready(backward=True) behind both."""
    from scenerf_amd.renderer import MLP_PARAM_NAMES, PackedMLP
    rcfg = RenderConfig.kitti(precision="bf16", n_pts_uni=64, n_pts_per_gaussian=16)
    state = synth.mlp_state(41, 2, out_scale=4.0)
    params = [state[n].to(DEV) for n in MLP_PARAM_NAMES]
    one = PackedMLP(params, 2, rcfg)
    side = torch.cuda.Stream()
    two = PackedMLP(params, 2, rcfg, pack_stream=side, defer=True, split=True)
    assert two._pending is not None and two._split
    two.gflat.fill_(float("nan"))            # (the second call zeroes the sink)
    torch.cuda.synchronize()
    two.wait_ready()
    assert two._pending is None and two._ready is None and two._ready_rest is not None
    two.wait_ready(backward=True)
    assert two._ready_rest is None
    torch.cuda.synchronize()
    assert torch.equal(one.act_buf.view(torch.int16), two.act_buf.view(torch.int16))
    assert torch.equal(one.f32_buf, two.f32_buf)
    assert not two.gflat.any()


def test_fill_zero_streams_zeroes_over_the_whole_buffer():
    """scenerf_hip_fill_zero (the accumulator fill a training session launches beside the head's forward): every byte zero for sizes that
    are not a multiple of the grid's stride, with few and with many workgroups; misaligned / ragged arguments are refused."""
    lib = _capi.load()
    for n, wg in ((4 * 1000 + 4, 3), (1 << 20, 128), (4 * 77777, 1024)):
        t = torch.full((n + 8,), float("nan"), device=DEV)
        _capi.check(lib.scenerf_hip_fill_zero(t.data_ptr(), n * 4, wg, _st()), "fill_zero")
        torch.cuda.synchronize()
        assert not t[:n].any() and bool(torch.isnan(t[n:]).all())
    t = torch.zeros(64, device=DEV)
    assert lib.scenerf_hip_fill_zero(t.data_ptr() + 4, 64, 4, _st()) != 0
    assert lib.scenerf_hip_fill_zero(t.data_ptr(), 60, 4, _st()) != 0
