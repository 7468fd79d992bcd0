"""Caller-side loss pieces (scenerf_amd/training.py): the sync-free restatements against values produced by the
reference's own functions (tests/golden/make_golden_loss.py).  CPU only."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden_loss import inputs   # noqa: E402  (input generator only; the reference import lives under __main__)

from scenerf_amd.training import TrainingMixin, depth_errors, sample_pix_features   # noqa: E402

G = np.load(os.path.join(HERE, "golden", "loss_side.npz"))


def _case(name, behind):
    d = {k: torch.from_numpy(v) for k, v in inputs(11 if not behind else 12, behind=behind).items()}
    col_src = sample_pix_features(d["pix"], d["img_s"])
    torch.testing.assert_close(col_src, torch.from_numpy(G[name + "/col_src"]), rtol=0, atol=0)   # utils.py:250-266
    orig = torch.randn
    torch.randn = lambda *a, **k: torch.zeros(*a, **{kk: vv for kk, vv in k.items() if kk in ("device", "dtype")})
    try:
        loss = TrainingMixin.compute_reprojection_loss(None, d["pix"], col_src, d["depth"], d["img_t"], torch.inverse(d["K"]), d["K"], d["T"])
    finally:
        torch.randn = orig
    return float(loss), int(G[name + "/n_valid"])


def test_reprojection_loss_all_rays_valid():
    loss, n_valid = _case("all_valid", False)
    assert n_valid == 300
    assert abs(loss - float(G["all_valid/loss_mean"])) < 1e-6


def test_reprojection_loss_masked_mean_equals_boolean_indexing():
    """268 of the 300 target points lie in front of the camera: mean over valid == sum(m * x) / sum(m) (no boolean indexing, no sync)."""
    loss, n_valid = _case("some_behind", True)
    assert 0 < n_valid < 300
    assert abs(loss - float(G["some_behind/loss_mean"])) < 1e-6


def test_depth_errors_match_reference_metrics():
    gt, pred = torch.from_numpy(G["depth/gt"]), torch.from_numpy(G["depth/pred"])
    got = torch.stack(depth_errors(gt, pred)).double().numpy()
    np.testing.assert_allclose(got, G["depth/metrics"], rtol=1e-6, atol=1e-9)      # loss/depth_metrics.py:3-24


def test_bundlefusion_forward_matches_the_reference_loop():
    """scenerf_bf.py:124-247 (BundleFusion batch layout, n_rays // grid**2 rays per source, x5 / x0.1 weights, depth metrics at the
    sampled pixels with eval_depth as the clamp): scenerf_amd's sync-free loop against the reference's own forward, both around the
    same deterministic fake renderer / encoder (tests/golden/bf_fakes.py, make_golden_bf_forward.py)."""
    from bf_fakes import FakeNetRgb, fake_batch, fake_render
    from scenerf_amd.model import SceneRFBundleFusion
    g = np.load(os.path.join(HERE, "golden", "bf_forward.npz"))
    m = SceneRFBundleFusion(som_sigma=2.0, img_size=(64, 48), n_rays=256, sample_grid_size=2, sphere_H=48, sphere_W=64, eval_depth=10)
    m.net_rgb = FakeNetRgb()
    m.render_rays_batch = lambda cam_K, T, x_rgb, ray_batch_size=None, sampled_pixels=None, **k: fake_render(sampled_pixels, T)
    logs = {}
    m.log = lambda key, val, **k: logs.setdefault(key, []).append(float(val))
    orig = torch.randn
    torch.randn = lambda *a, **k: torch.zeros(*a, **{kk: vv for kk, vv in k.items() if kk in ("device", "dtype")})
    try:
        torch.manual_seed(5)
        out = m.forward(fake_batch(seed=3), "train")
    finally:
        torch.randn = orig
    assert abs(float(out["total_loss"]) - float(g["total_loss"])) < 2e-6
    keys = [k[4:] for k in g.files if k.startswith("log/")]
    assert sorted(keys) == sorted(logs), (sorted(set(keys) ^ set(logs)))
    for k in keys:
        np.testing.assert_allclose(np.asarray(logs[k]), g["log/" + k], rtol=2e-5, atol=2e-6, err_msg=k)
    with torch.no_grad():
        m.smooth_loss_weight = 0.1
    try:
        m.forward(fake_batch(seed=3), "train")
        raise AssertionError("smooth_loss_weight > 0 must fail like the reference (compute_smooth_depth_loss does not exist)")
    except NotImplementedError:
        pass


def test_kitti_forward_matches_the_reference_loop():
    """scenerf.py:119-241 (per-sample intrinsics, n_rays rays per source, x1 / x0.01 weights, lidar depth metrics from a second
    render): scenerf_amd's sync-free loop -- metric render under no_grad -- against the reference's own forward around the same
    fake renderer / encoder."""
    from bf_fakes import FakeNetRgb, fake_batch_kitti, fake_render
    from scenerf_amd.model import SceneRF
    g = np.load(os.path.join(HERE, "golden", "kitti_forward.npz"))
    m = SceneRF(som_sigma=2.0, img_size=(64, 48), n_rays=200, sphere_H=48, sphere_W=64)
    m.net_rgb = FakeNetRgb()
    m.render_rays_batch = lambda cam_K, T, x_rgb, ray_batch_size=None, sampled_pixels=None, **k: fake_render(sampled_pixels, T)
    logs = {}
    m.log = lambda key, val, **k: logs.setdefault(key, []).append(float(val))
    orig = torch.randn
    torch.randn = lambda *a, **k: torch.zeros(*a, **{kk: vv for kk, vv in k.items() if kk in ("device", "dtype")})
    try:
        torch.manual_seed(6)
        out = m.forward(fake_batch_kitti(seed=4), "train")
    finally:
        torch.randn = orig
    assert abs(float(out["total_loss"]) - float(g["total_loss"])) < 2e-6
    keys = [k[4:] for k in g.files if k.startswith("log/")]
    assert sorted(keys) == sorted(logs), (sorted(set(keys) ^ set(logs)))
    for k in keys:
        np.testing.assert_allclose(np.asarray(logs[k]), g["log/" + k], rtol=2e-5, atol=2e-6, err_msg=k)


def test_reprojection_loss_has_no_nan_gradient_for_points_on_the_camera_plane():
    """A target point with z == 0 is masked out of the loss; its (unselected) perspective division must not leak 0/0 = NaN into
    the gradient of the rendered depth (torch.where back-propagates through both branches)."""
    d = {k: torch.from_numpy(v) for k, v in inputs(11, behind=False).items()}
    col_src = sample_pix_features(d["pix"], d["img_s"])
    T = d["T"].clone()
    T[2, :] = 0.0            # every target point lands on z == 0
    depth = d["depth"].clone().requires_grad_(True)
    loss = TrainingMixin.compute_reprojection_loss(None, d["pix"], col_src, depth, d["img_t"], torch.inverse(d["K"]), d["K"], T)
    loss.backward()
    assert float(loss) == 0.0 and bool(torch.isfinite(depth.grad).all())


def test_masked_depth_metrics_are_skipped_when_nothing_is_valid():
    """scenerf_bf.py:204-205: a source frame without a valid depth logs nothing (instead of all-zero metrics that bias the epoch means)."""
    logs = []

    class M(TrainingMixin):
        def log(self, k, v, **kw):
            logs.append(k)
    m = M()
    gt, pred = torch.zeros(10), torch.ones(10)
    m.evaluate_depth("val", gt, pred, mask=gt > 0)
    assert logs == []
    m.evaluate_depth("val", gt + 1, pred, mask=(gt + 1) > 0)
    assert len(logs) == 7
