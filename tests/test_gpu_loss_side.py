"""SURVEY section 8f-1 on the GPU: the fused loss-side kernel (csrc/loss.hip behind scenerf_amd.loss_side.loss_side) against
(a) the values the reference's own functions produced (tests/golden/loss_side.npz, minted by make_golden_loss.py) and
(b) the stock-PyTorch restatement in scenerf_amd/training.py (itself pinned on those goldens by tests/test_loss_side.py) run through
torch autograd on the same device: loss_color, the reprojection loss, and the gradients w.r.t. the rendered colour and depth."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden_loss import inputs   # noqa: E402

from scenerf_amd.training import TrainingMixin, sample_pix_features   # noqa: E402

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(HERE, "golden", "loss_side.npz"))
DEV = "cuda:0"


def _torch_reference(d, color, noise):
    col_src = sample_pix_features(d["pix"], d["img_s"])
    loss_color = torch.abs(color - col_src.T)
    orig = torch.randn
    torch.randn = lambda *a, **k: (noise / 0.00001)   # compute_reprojection_loss multiplies its draw by 1e-5
    try:
        loss_rep = TrainingMixin.compute_reprojection_loss(None, d["pix"], col_src, d["depth"], d["img_t"], torch.inverse(d["K"]), d["K"], d["T"])
    finally:
        torch.randn = orig
    return loss_color, loss_rep, col_src


@pytest.mark.parametrize("name,behind,seed", [("all_valid", False, 11), ("some_behind", True, 12)])
def test_fused_loss_side_matches_reference_values_and_torch_autograd(name, behind, seed):
    from scenerf_amd.loss_side import loss_side
    d = {k: torch.from_numpy(v).to(DEV) for k, v in inputs(seed, behind=behind).items()}
    R = d["pix"].shape[0]
    gen = torch.Generator().manual_seed(seed)
    color0 = torch.rand(R, 3, generator=gen).to(DEV)
    zero = torch.zeros(R, device=DEV)
    # (a) the reference's numbers (no noise in the golden run)
    lc, lr = loss_side(color0, d["depth"], d["pix"], d["img_s"], d["img_t"], d["K"], torch.inverse(d["K"]), d["T"], None)
    assert abs(float(lr) - float(G[name + "/loss_mean"])) < 2e-6, (float(lr), float(G[name + "/loss_mean"]))
    col_src_ref = torch.from_numpy(G[name + "/col_src"]).to(DEV)
    torch.testing.assert_close(lc, torch.abs(color0 - col_src_ref.T), rtol=0, atol=2e-6)
    # (b) values and gradients against torch autograd, with noise on the identity term
    noise = (torch.randn(R, generator=gen) * 0.00001).to(DEV)
    w_lc = torch.rand(R, 3, generator=gen).to(DEV)
    outs = {}
    for kind in ("fused", "torch"):
        color = color0.clone().requires_grad_(True)
        depth = d["depth"].clone().requires_grad_(True)
        dd = dict(d, depth=depth)
        if kind == "fused":
            lc, lr = loss_side(color, depth, d["pix"], d["img_s"], d["img_t"], d["K"], torch.inverse(d["K"]), d["T"], noise)
        else:
            lc, lr, _ = _torch_reference(dd, color, noise)
        ((lc * w_lc).sum() + 3.0 * lr).backward()
        outs[kind] = (lc.detach(), lr.detach(), color.grad.clone(), depth.grad.clone())
    (lc_a, lr_a, gc_a, gd_a), (lc_b, lr_b, gc_b, gd_b) = outs["fused"], outs["torch"]
    # (torch's GPU grid_sample and the kernel round their bilinear weights differently: measured 3.3e-6 on values in [0, 1];
    # against the reference's CPU numbers above the kernel is within 2e-6)
    torch.testing.assert_close(lc_a, lc_b, rtol=0, atol=8e-6)
    assert abs(float(lr_a) - float(lr_b)) < 8e-6
    torch.testing.assert_close(gc_a, gc_b, rtol=0, atol=1e-6)
    assert float(gd_b.abs().max()) > 0, "the case must exercise the depth gradient"
    scale = float(gd_b.abs().max())
    print("depth gradient: max |diff| %.3e of scale %.3e" % (float((gd_a - gd_b).abs().max()), scale))
    assert float((gd_a - gd_b).abs().max()) <= 2e-4 * scale, (float((gd_a - gd_b).abs().max()), scale)


def test_fused_loss_side_refuses_cpu_tensors():
    from scenerf_amd.loss_side import loss_side
    d = {k: torch.from_numpy(v) for k, v in inputs(11, behind=False).items()}
    with pytest.raises(RuntimeError, match="GPU"):
        loss_side(torch.rand(300, 3), d["depth"], d["pix"], d["img_s"], d["img_t"], d["K"], torch.inverse(d["K"]), d["T"], None)


@pytest.mark.parametrize("behind,seed,R_rep", [(False, 21, 1), (True, 22, 1), (True, 23, 5)])
def test_source_loss_matches_the_torch_assembly(behind, seed, R_rep):
    """scenerf_amd.loss_side.source_loss (the whole loss of one source frame, one launch each way) against the reference's assembly
    written with the stock-PyTorch pieces of scenerf_amd/training.py (pinned on the reference's own functions by tests/test_loss_side.py):
    total, every logged term, and the gradients w.r.t. colour, depth, loss_kl and the gaussian means.  R_rep > 1: more than 1,024 rays
    (the per-block partial sums and the finishing launch)."""
    from scenerf_amd.loss_side import source_loss
    d = {k: torch.from_numpy(v).to(DEV) for k, v in inputs(seed, behind=behind).items()}
    if R_rep > 1:
        d["pix"], d["depth"] = d["pix"].repeat(R_rep, 1), d["depth"].repeat(R_rep) * (1 + 0.01 * torch.arange(d["depth"].shape[0] * R_rep, device=DEV) / 100)
    R, Gn = d["pix"].shape[0], 4
    gen = torch.Generator().manual_seed(seed)
    color0 = torch.rand(R, 3, generator=gen).to(DEV)
    kl0 = torch.rand(R, generator=gen).to(DEV)
    gm0 = (torch.rand(R, Gn, generator=gen).to(DEV) * 2 - 1) * 3 + d["depth"][:, None]
    gs0, sv0 = torch.rand(R, Gn, generator=gen).to(DEV) + 1.5, torch.rand(R, Gn, generator=gen).to(DEV)
    noise = torch.randn(R, generator=gen).to(DEV)
    w = dict(reproj_weight=5.0, color_weight=1.0, dist2closest_weight=0.1)
    outs = {}
    for kind in ("fused", "torch"):
        color, depth = color0.clone().requires_grad_(True), d["depth"].clone().requires_grad_(True)
        kl, gm = kl0.clone().requires_grad_(True), gm0.clone().requires_grad_(True)
        out = {"color": color, "depth": depth, "loss_kl": kl, "gaussian_means": gm, "gaussian_stds": gs0, "som_vars": sv0}
        if kind == "fused":
            total, terms = source_loss(out, d["pix"], d["img_s"], d["img_t"], d["K"], torch.inverse(d["K"]), d["T"], noise=noise, noise_scale=1e-5, **w)
            terms = terms.detach()
        else:
            lc, lr, _ = _torch_reference(dict(d, depth=depth), color, noise * 0.00001)
            diff = torch.abs(gm - depth.unsqueeze(-1).detach())
            min_diff, gi = torch.min(diff, dim=1)
            total = lr * w["reproj_weight"] + lc.mean() * w["color_weight"] + kl.mean() + min_diff.mean() * w["dist2closest_weight"]
            terms = torch.stack([total.detach(), lr.detach(), lc.mean().detach(), kl.mean().detach(), min_diff.mean().detach(),
                                 torch.gather(sv0, 1, gi[:, None]).mean(), torch.gather(gs0, 1, gi[:, None]).mean()])
        (total * 1.7).backward()
        outs[kind] = (float(total), terms[:7].cpu(), color.grad.clone(), depth.grad.clone(), kl.grad.clone(), gm.grad.clone())
    a, b = outs["fused"], outs["torch"]
    assert abs(a[0] - b[0]) <= 2e-5 * (1 + abs(b[0])), (a[0], b[0])
    torch.testing.assert_close(a[1], b[1], rtol=2e-5, atol=8e-6)
    torch.testing.assert_close(a[2], b[2], rtol=1e-5, atol=1e-9)       # colour: sign / (3 R)
    scale = float(b[3].abs().max())
    assert scale > 0 and float((a[3] - b[3]).abs().max()) <= 2e-4 * scale
    torch.testing.assert_close(a[4], b[4], rtol=1e-6, atol=0)
    torch.testing.assert_close(a[5], b[5], rtol=1e-6, atol=0)
    # reproducible run to run (fixed summation order, no atomics)
    out = {"color": color0, "depth": d["depth"], "loss_kl": kl0, "gaussian_means": gm0, "gaussian_stds": gs0, "som_vars": sv0}
    t1 = source_loss(out, d["pix"], d["img_s"], d["img_t"], d["K"], torch.inverse(d["K"]), d["T"], noise=noise, **w)[1].clone()
    t2 = source_loss(out, d["pix"], d["img_s"], d["img_t"], d["K"], torch.inverse(d["K"]), d["T"], noise=noise, **w)[1].clone()
    assert torch.equal(t1, t2)


def test_source_loss_in_kernel_noise_is_standard_normal_and_advances():
    """rng_state = [seed, calls]: the tie-breaking noise of scenerf.py:378 made inside the kernel.  With a huge noise scale s the ray term
    is min(l_rep, l_id + s n) ~ s min(0, n), whose mean over rays is -s / sqrt(2 pi) for n ~ N(0, 1): a moment check of the generator
    through the only place its values go.  The launch advances the call counter (a replayed graph draws fresh noise); the same state gives
    the same numbers."""
    import math
    from scenerf_amd.loss_side import make_rng_state, source_loss
    d = {k: torch.from_numpy(v).to(DEV) for k, v in inputs(31, n=4000, behind=False).items()}
    R = d["pix"].shape[0]
    gen = torch.Generator().manual_seed(31)
    out = {"color": torch.rand(R, 3, generator=gen).to(DEV), "depth": d["depth"], "loss_kl": torch.rand(R, generator=gen).to(DEV),
           "gaussian_means": torch.rand(R, 4, generator=gen).to(DEV) * 20, "gaussian_stds": None, "som_vars": None}
    args = (d["pix"], d["img_s"], d["img_t"], d["K"], torch.inverse(d["K"]), d["T"])
    st = make_rng_state(DEV, seed=1234)
    assert st.tolist() == [1234, 0]
    s_big = 1e4
    t1 = source_loss(out, *args, noise=None, noise_scale=s_big, rng_state=st)[1].clone()
    assert st.tolist() == [1234, 1]
    t2 = source_loss(out, *args, noise=None, noise_scale=s_big, rng_state=st)[1].clone()
    assert st.tolist() == [1234, 2] and float(t1[1]) != float(t2[1])            # fresh noise per call
    for t in (t1, t2):
        assert abs(float(t[1]) / s_big + 1 / math.sqrt(2 * math.pi)) < 0.04, float(t[1]) / s_big      # E[min(0, n)] = -0.3989 (4,000 rays: SE 0.009)
    st.copy_(torch.tensor([1234, 0]))
    t1b = source_loss(out, *args, noise=None, noise_scale=s_big, rng_state=st)[1]
    assert torch.equal(t1, t1b)                                                 # same state, same values
    # scale 0 == no noise at all
    a = source_loss(out, *args, noise=None, noise_scale=0.0, rng_state=st)[1]
    b = source_loss(out, *args, noise=None)[1]
    assert torch.equal(a, b)


@pytest.mark.gpu
def test_depth_errors_kernel_matches_the_reference_metrics_and_the_host_version():
    """scenerf_hip_depth_errors (one launch) against (i) the metrics the reference's compute_depth_errors returned for the golden pair
    (loss/depth_metrics.py:3-24, tests/golden/loss_side.npz) and (ii) training.depth_errors' elementwise version on the CPU for clamped
    predictions, masks (scenerf_bf.py:201-205), an empty mask and a 300,000-entry case (one block strides over it)."""
    from scenerf_amd.training import depth_errors
    gt, pred = torch.from_numpy(G["depth/gt"]), torch.from_numpy(G["depth/pred"])
    got = torch.stack(depth_errors(gt.to(DEV), pred.to(DEV))).double().cpu().numpy()
    np.testing.assert_allclose(got, G["depth/metrics"], rtol=2e-6, atol=1e-9)
    g = torch.Generator().manual_seed(3)
    for n, masked, md in ((1200, False, 80.0), (1200, True, 10.0), (7, True, 80.0), (300000, True, 80.0), (64, "empty", 80.0)):
        gt = torch.rand(n, generator=g) * 90 + 0.5
        pred = gt * (1 + 0.3 * torch.randn(n, generator=g)) + 0.01
        pred[::17] = 200.0
        pred[::23] = -1.0
        mask = None if not masked else (torch.zeros(n, dtype=torch.bool) if masked == "empty" else torch.rand(n, generator=g) > 0.4)
        want = torch.stack(depth_errors(gt, pred, max_depth=md, mask=mask)).double().numpy()
        got = torch.stack(depth_errors(gt.to(DEV), pred.to(DEV), max_depth=md, mask=None if mask is None else mask.to(DEV))).double().cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-7, err_msg=str((n, masked, md)))     # (fp32 means on the host, double sums in the kernel)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["kitti", "bundlefusion"])
def test_trainer_forward_on_the_gpu_matches_the_reference_loop(variant):
    """The reference's own ``forward`` (scenerf.py:119-241 / scenerf_bf.py:124-247) run around a deterministic fake renderer and encoder
    minted these totals and logged values (tests/golden/make_golden_{kitti,bf}_forward.py).  On CUDA tensors scenerf_amd's loop takes the
    paths the CPU test (tests/test_loss_side.py) does not: the whole per-source loss in one kernel each way, the logged terms as one
    vector, the depth metrics in one launch, the per-image scope (pixel subsets drawn up front stays off: host draw like the reference)
    and, for KITTI, the metric-only renders on a stream of their own."""
    from bf_fakes import FakeNetRgb, fake_batch, fake_batch_kitti, fake_render
    from scenerf_amd.model import SceneRF, SceneRFBundleFusion

    def to_dev(x):
        if torch.is_tensor(x):
            return x.to(DEV)
        if isinstance(x, np.ndarray):
            return torch.from_numpy(x).to(DEV)
        if isinstance(x, (list, tuple)):
            return [to_dev(v) for v in x]
        return x

    if variant == "kitti":
        g = np.load(os.path.join(HERE, "golden", "kitti_forward.npz"))
        m = SceneRF(som_sigma=2.0, img_size=(64, 48), n_rays=200, sphere_H=48, sphere_W=64)
        batch, seed = fake_batch_kitti(seed=4), 6
    else:
        g = np.load(os.path.join(HERE, "golden", "bf_forward.npz"))
        m = SceneRFBundleFusion(som_sigma=2.0, img_size=(64, 48), n_rays=256, sample_grid_size=2, sphere_H=48, sphere_W=64, eval_depth=10)
        batch, seed = fake_batch(seed=3), 5
    m = m.to(DEV)
    batch = {k: to_dev(v) for k, v in batch.items()}

    class Enc(torch.nn.Module):
        def forward(self, img, pix=None, pix_sphere=None):
            return {k: v.to(DEV) for k, v in FakeNetRgb()(img.cpu()).items()}

    m.net_rgb = Enc()
    m.render_rays_batch = lambda cam_K, T, x_rgb, ray_batch_size=None, sampled_pixels=None, **k: {
        kk: vv.to(DEV) for kk, vv in fake_render(sampled_pixels.cpu(), T.cpu()).items()}
    m.fused_loss_noise = "torch"            # the tie-breaking noise through torch.randn, which the golden run replaced by zeros
    logs = {}
    m.log = lambda key, val, **k: logs.setdefault(key, []).append(val.detach().clone() if torch.is_tensor(val) else val)
    orig = torch.randn
    torch.randn = lambda *a, **k: torch.zeros(*a, **{kk: vv for kk, vv in k.items() if kk in ("device", "dtype")})
    try:
        torch.manual_seed(seed)
        out = m.forward(batch, "train")
    finally:
        torch.randn = orig
    torch.cuda.synchronize()
    assert abs(float(out["total_loss"]) - float(g["total_loss"])) < 5e-6 * (1 + abs(float(g["total_loss"])))
    keys = [k[4:] for k in g.files if k.startswith("log/")]
    assert sorted(keys) == sorted(logs), (sorted(set(keys) ^ set(logs)))
    for k in keys:
        np.testing.assert_allclose(np.asarray([float(v) for v in logs[k]]), g["log/" + k], rtol=3e-5, atol=3e-6, err_msg=k)
