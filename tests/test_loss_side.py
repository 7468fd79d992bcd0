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
