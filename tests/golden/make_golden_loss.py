"""Golden vectors for the caller-side loss pieces from the reference itself: utils.sample_pix_features,
SceneRF.compute_reprojection_loss (scenerf.py:349-386) and loss/depth_metrics.compute_depth_errors.  Build container only.
The identity-reprojection noise (torch.randn * 1e-5) is replaced by zeros so the two implementations can be compared when some
rays are masked out (they draw differently shaped noise)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _install_reference, REF   # noqa: E402


def inputs(seed, n=300, H=37, W=61, behind=False):
    rng = np.random.Generator(np.random.PCG64(seed))
    K = np.array([[40.0, 0, 30.0], [0, 40.0, 18.0], [0, 0, 1]], dtype=np.float32)
    pix = np.stack([rng.uniform(0, W - 1, n), rng.uniform(0, H - 1, n)], 1).astype(np.float32)
    img_s = rng.random((3, H, W), dtype=np.float32)
    img_t = rng.random((3, H, W), dtype=np.float32)
    depth = rng.uniform(1.0, 20.0, n).astype(np.float32)
    T = np.eye(4, dtype=np.float32)
    ang = 0.05
    T[:3, :3] = [[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]]
    T[:3, 3] = [0.3, -0.05, -3.0 if behind else 0.4]     # behind: the source camera sits 3 m in front -> near points project behind
    return dict(K=K, pix=pix, img_s=img_s, img_t=img_t, depth=depth, T=T)


if __name__ == "__main__":
    assert os.path.isdir(REF)
    _install_reference()
    from scenerf.models import utils as U
    from scenerf.models.scenerf import SceneRF
    from scenerf.loss.depth_metrics import compute_depth_errors
    blob = {}
    orig_randn = torch.randn
    torch.randn = lambda *a, **k: torch.zeros(*a, **{kk: vv for kk, vv in k.items() if kk in ("device", "dtype")})

    class Dev:   # the method only reads self.device
        device = torch.device("cpu")
    for name, behind in (("all_valid", False), ("some_behind", True)):
        d = {k: torch.from_numpy(v) for k, v in inputs(11 if not behind else 12, behind=behind).items()}
        col_src = U.sample_pix_features(d["pix"], d["img_s"])
        loss = SceneRF.compute_reprojection_loss(Dev(), d["pix"], col_src, d["depth"], d["img_t"], torch.inverse(d["K"]), d["K"], d["T"])
        blob[name + "/col_src"] = col_src.numpy()
        blob[name + "/loss_mean"] = np.float64(loss.mean().item())
        blob[name + "/n_valid"] = np.int64(loss.numel())
    torch.randn = orig_randn
    rng = np.random.Generator(np.random.PCG64(13))
    gt = rng.uniform(0.5, 70, 500); pred = gt * rng.uniform(0.6, 1.5, 500); pred[:5] = 1e-5; pred[5:9] = 200.0
    blob["depth/gt"], blob["depth/pred"] = gt, pred.copy()
    blob["depth/metrics"] = np.asarray(compute_depth_errors(gt, pred.copy()), dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "loss_side.npz"), **blob)
    print({k: (v if np.ndim(v) == 0 else np.shape(v)) for k, v in blob.items()})
