"""Image -> sphere resampling of the encoder levels (reference unet2d_sphere.py:138-165).
CPU: the numpy oracle against what the reference's own ``DecoderSphere.get_sphere_feature`` produced (tests/golden/
make_golden_sphere.py: full data on a 1/10-size geometry, the reference's scattered maps for all six levels at KITTI size).
GPU: the HIP kernels through the C ABI against the oracle (bit-exact: maps, forward, gather-form backward), against the golden
vectors, against torch's own grid_sample on the same GPU, and size-independent properties at the full KITTI sizes."""
import os

import numpy as np
import pytest
import torch

import sphere_oracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sphere_resample.npz")
LEVELS = (1, 2, 4, 8, 16, 32)


def _grid(img_W, img_H):
    ys, xs = np.meshgrid(np.arange(img_H), np.arange(img_W), indexing="ij")
    return np.stack([xs.reshape(-1), ys.reshape(-1)], 1).astype(np.float32)


def _small(g):
    img_W, img_H, out_W, out_H = (int(v) for v in g["small/dims"])
    return _grid(img_W, img_H), g["small/pix_sphere"].astype(np.int64), out_W, out_H


def _kitti(g):
    ps = np.cumsum(g["kitti/pix_sphere_d"].astype(np.int64), axis=1).reshape(-1, 2)
    return _grid(1220, 370), ps, 1500, 452


def _ref_src(g, s):
    return np.cumsum(g[f"kitti/s{s}/src_d"].astype(np.int64), axis=1).astype(np.int32)


def test_oracle_maps_equal_the_reference_maps_at_kitti_size():
    g = np.load(GOLD)
    pix, ps, out_W, out_H = _kitti(g)
    for s in LEVELS:
        ow, oh = orc.scaled_dims(out_W, out_H, s)
        src = orc.build_map(pix, ps, s, ow, oh)
        assert src.shape == (oh, ow) and np.array_equal(src, _ref_src(g, s)), "level %d" % s
        w, h = (int(v) for v in g[f"kitti/s{s}/plane"])
        ones = orc.resample_forward(np.ones((1, 1, h, w), np.float32), src)[0, 0]
        assert np.array_equal(np.rint(ones * 4).astype(np.uint8), g[f"kitti/s{s}/ones_x4"])
    assert orc.scaled_dims(1500, 452, 8) == (188, 56)          # Python round: half to even
    # the scatter really has duplicates (1220 pixel columns land on ~990 sphere columns): the winner rule matters
    u = np.rint(ps[:, 0].astype(np.float32)).astype(np.int64); v = np.rint(ps[:, 1].astype(np.float32)).astype(np.int64)
    assert np.unique(v * 1500 + u).size < 0.7 * ps.shape[0]


def test_oracle_forward_backward_match_the_reference():
    g = np.load(GOLD)
    pix, ps, out_W, out_H = _small(g)
    for s in (1, 2, 4, 8):
        x = g[f"small/s{s}/x"]
        ow, oh = orc.scaled_dims(out_W, out_H, s)
        src = orc.build_map(pix, ps, s, ow, oh)
        out = orc.resample_forward(x, src)
        assert np.abs(out - g[f"small/s{s}/out"]).max() <= 2.5e-7          # fp32 rounding of a 4-term sum of O(1) values
        dx = orc.resample_backward(g[f"small/s{s}/g"], src, x.shape[2], x.shape[3])
        assert np.abs(dx - g[f"small/s{s}/dx"]).max() <= 1e-6


def _odd_map(rng, H, W, oh, ow):
    """A synthetic map with everything the geometry maps do not have: entries on every border, one past the plane, far outside
    (no taps), many cells per pixel and empty cells."""
    sx = rng.integers(0, W + 1, size=(oh, ow)); sy = rng.integers(0, H + 1, size=(oh, ow))
    src = ((sy << 16) | sx).astype(np.int32)
    src[rng.random((oh, ow)) < 0.3] = -1
    src[0, 0] = ((H + 5) << 16) | 3          # far outside
    src[0, 1] = (2 << 16) | (W + 7)
    src[1, :4] = (0 << 16) | 0               # a pile-up on the corner pixel
    return src


def test_gather_form_backward_is_the_adjoint():
    rng = np.random.Generator(np.random.PCG64(5))
    H, W, oh, ow = 9, 13, 17, 21
    src = _odd_map(rng, H, W, oh, ow)
    x = rng.standard_normal((1, 3, H, W), dtype=np.float32)
    dout = rng.standard_normal((1, 3, oh, ow), dtype=np.float32)
    row_ptr, cells = orc.csr_of(src, H, W)
    assert row_ptr.shape == ((H + 1) * (W + 1) + 1,) and row_ptr[-1] == cells.size < (src >= 0).sum()   # the far-outside cells dropped
    dx_a = orc.resample_backward(dout, src, H, W)
    dx_g = orc.resample_backward_gather(dout, row_ptr, cells, H, W)
    assert np.abs(dx_g - dx_a).max() <= 2e-6
    lhs = float((orc.resample_forward(x, src).astype(np.float64) * dout).sum())
    assert abs(lhs - float((x.astype(np.float64) * dx_a).sum())) <= 1e-5 * max(1.0, abs(lhs))


def test_host_side_fails_loudly_without_a_gpu_tensor():
    from scenerf_amd.sphere import SphereResampler, scaled_dims
    assert scaled_dims(1500, 452, 8) == (188, 56) and scaled_dims(1500, 452, 32) == (47, 14)
    rs = SphereResampler(150, 45)
    pix = torch.from_numpy(_grid(12, 4)); ps = torch.zeros(48, 2, dtype=torch.long)
    with pytest.raises(RuntimeError, match="GPU only|not built"):
        rs.get_sphere_feature(torch.zeros(1, 1, 4, 12), pix, ps, 1)


# ------------------------------------------------------------------------------------------------------------------ GPU

def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
def test_hip_map_build_is_bit_exact():
    from scenerf_amd.sphere import build_map, scaled_dims
    g = np.load(GOLD)
    for geom, levels in ((_kitti(g), LEVELS), (_small(g), (1, 2, 4, 8))):
        pix, ps, out_W, out_H = geom
        dpix, dps = _dev(pix), _dev(ps)
        for s in levels:
            ow, oh = scaled_dims(out_W, out_H, s)
            m = build_map(dpix, dps, s, ow, oh, 1, 1)
            assert np.array_equal(m.src.cpu().numpy(), orc.build_map(pix, ps, s, ow, oh)), "level %d" % s
    # the reference's own maps, directly
    pix, ps, out_W, out_H = _kitti(g)
    for s in LEVELS:
        ow, oh = scaled_dims(out_W, out_H, s)
        assert np.array_equal(build_map(_dev(pix), _dev(ps), s, ow, oh, 1, 1).src.cpu().numpy(), _ref_src(g, s))


@pytest.mark.gpu
def test_hip_forward_backward_match_oracle_and_reference():
    from scenerf_amd.sphere import SphereResampler
    g = np.load(GOLD)
    pix, ps, out_W, out_H = _small(g)
    rs = SphereResampler(out_W, out_H)
    dpix, dps = _dev(pix), _dev(ps)
    for s in (1, 2, 4, 8):
        x = g[f"small/s{s}/x"]; gout = g[f"small/s{s}/g"]
        xt = _dev(x).requires_grad_(True)
        out = rs.get_sphere_feature(xt, dpix, dps, s)
        out.backward(_dev(gout))
        ow, oh = orc.scaled_dims(out_W, out_H, s)
        src = orc.build_map(pix, ps, s, ow, oh)
        assert np.array_equal(out.detach().cpu().numpy(), orc.resample_forward(x, src))            # same operation sequence
        row_ptr, cells = orc.csr_of(src, x.shape[2], x.shape[3])
        m = rs.map_for(dpix, dps, s, x.shape[2], x.shape[3])
        assert np.array_equal(m.csr()[0].cpu().numpy(), row_ptr) and np.array_equal(m.csr()[1].cpu().numpy(), cells)
        assert np.array_equal(xt.grad.cpu().numpy(), orc.resample_backward_gather(gout, row_ptr, cells, x.shape[2], x.shape[3]))
        assert np.abs(out.detach().cpu().numpy() - g[f"small/s{s}/out"]).max() <= 2.5e-7         # the reference's numbers
        assert np.abs(xt.grad.cpu().numpy() - g[f"small/s{s}/dx"]).max() <= 2e-6


@pytest.mark.gpu
def test_hip_odd_maps_and_plane_tails():
    """Synthetic maps (borders, one-past-the-plane, pile-ups, empties) and plane counts that are not a multiple of the four planes
    a thread handles."""
    from scenerf_amd import _capi
    from scenerf_amd.sphere import SphereMap, resample
    rng = np.random.Generator(np.random.PCG64(6))
    for (H, W, oh, ow, B, C) in ((9, 13, 17, 21, 1, 3), (33, 70, 40, 300, 2, 5), (12, 39, 14, 47, 1, 9)):
        src = _odd_map(rng, H, W, oh, ow)
        x = rng.standard_normal((B, C, H, W), dtype=np.float32)
        dout = rng.standard_normal((B, C, oh, ow), dtype=np.float32)
        m = SphereMap(_dev(src), H, W)
        xt = _dev(x).requires_grad_(True)
        out = resample(xt, m)
        out.backward(_dev(dout))
        assert np.array_equal(out.detach().cpu().numpy(), orc.resample_forward(x, src))
        row_ptr, cells = orc.csr_of(src, H, W)
        assert np.array_equal(xt.grad.cpu().numpy(), orc.resample_backward_gather(dout, row_ptr, cells, H, W))
        # channels-last sphere side (SphereResampler(layout="hwc"), what renderer.HWC reads in place): the same numbers, permuted
        xh = _dev(x).requires_grad_(True)
        outh = resample(xh, m, True)
        assert outh.shape == (B, oh, ow, C) and torch.equal(outh.detach(), out.detach().permute(0, 2, 3, 1))
        outh.backward(_dev(dout).permute(0, 2, 3, 1).contiguous())
        assert torch.equal(xh.grad, xt.grad)
    with pytest.raises(RuntimeError, match="float32 CUDA"):
        resample(torch.zeros(1, 1, 9, 13, dtype=torch.float64).cuda(), m)
    assert _capi.load().scenerf_hip_sphere_resample_forward(None, 1, 4, 4, None, 4, 4, None, None) != 0     # NULL arguments are refused


@pytest.mark.gpu
def test_hip_against_torch_grid_sample_at_kitti_size():
    """Every level the decoder resamples, at its real size and channel count, against the reference's own formulation run with
    torch on the same GPU: the grid is rebuilt from OUR (deterministic) map exactly as unet2d_sphere.py:149-154 builds it, then
    F.grid_sample forward and backward.  Also: adjoint identity, determinism of the backward pass, map caching."""
    import torch.nn.functional as F
    from scenerf_amd.sphere import SphereResampler
    g = np.load(GOLD)
    pix, ps, out_W, out_H = _kitti(g)
    dpix, dps = _dev(pix), _dev(ps)
    rs = SphereResampler(out_W, out_H)
    gen = torch.Generator(device="cuda").manual_seed(3)
    for s, C in ((1, 3), (2, 32), (4, 48), (8, 80), (16, 224), (32, 2560)):
        w, h = (int(v) for v in g[f"kitti/s{s}/plane"])
        x = torch.randn(1, C, h, w, device="cuda", generator=gen).requires_grad_(True)
        out = rs.get_sphere_feature(x, dpix, dps, s)
        dout = torch.randn(out.shape, device="cuda", generator=gen)
        out.backward(dout)
        m = rs.map_for(dpix, dps, s, h, w)
        assert rs.map_for(dpix, dps, s, h, w) is m                                  # cached: no second scatter
        # the reference formulation on our map
        src = m.src.T.reshape(-1)                                                    # (out_W, out_H) order like the reference's map
        mx = torch.where(src >= 0, (src & 0xFFFF).float(), torch.full_like(src, -10, dtype=torch.float32))
        my = torch.where(src >= 0, (src >> 16).float(), torch.full_like(src, -10, dtype=torch.float32))
        grid = torch.stack([mx / w, my / h], 1) * 2 - 1
        x2 = x.detach().clone().requires_grad_(True)
        ref = F.grid_sample(x2, grid.reshape(1, 1, -1, 2), align_corners=False, mode="bilinear")
        ref = ref.reshape(1, C, m.out_w, m.out_h).permute(0, 1, 3, 2)
        ref.backward(dout)
        assert tuple(out.shape) == tuple(ref.shape)
        # torch's GPU kernel contracts ((g + 1) * W - 1) / 2 into an fma: its sample point differs from the eager-CPU sequence (which the
        # oracle and the HIP kernel follow bit for bit) by up to half an ulp of 2 W, and so do the bilinear weights -- 6e-5 at W = 1220.
        tol = max(2e-6, 3.0 * w * 2.0 ** -23 * float(x.detach().abs().max()))
        ef, eb = (out - ref).abs().max().item(), (x.grad - x2.grad).abs().max().item()
        assert ef <= tol, "level %d forward: %g > %g" % (s, ef, tol)
        assert eb <= 6 * tol, "level %d backward: %g > %g" % (s, eb, 6 * tol)    # up to ~6 cells per pixel; torch sums them with atomics
        lhs = (out.double() * dout.double()).sum().item(); rhs = (x.detach().double() * x.grad.double()).sum().item()
        assert abs(lhs - rhs) <= 1e-6 * max(1.0, abs(lhs)), "level %d adjoint" % s
        g1 = x.grad.clone(); x.grad = None
        rs.get_sphere_feature(x, dpix, dps, s).backward(dout)
        assert torch.equal(g1, x.grad), "level %d: backward must be deterministic" % s
