"""Golden values for the KITTI training forward (loss assembly, scenerf/models/scenerf.py:119-241) from the reference itself, the
same way as make_golden_bf_forward.py: the reference's ``SceneRF.forward`` on the CPU around the fake renderer / encoder of
bf_fakes.py, identity-reprojection noise replaced by zeros.  Build container only."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _install_reference, REF   # noqa: E402
from bf_fakes import FakeNetRgb, fake_batch_kitti, fake_render   # noqa: E402

if __name__ == "__main__":
    assert os.path.isdir(REF)
    _install_reference()
    from scenerf.models.scenerf import SceneRF
    m = SceneRF(som_sigma=2.0, img_size=(64, 48), n_rays=200, sphere_H=48, sphere_W=64)
    m.net_rgb = FakeNetRgb()
    m.render_rays_batch = lambda cam_K, T, x_rgb, ray_batch_size=None, sampled_pixels=None, **k: fake_render(sampled_pixels, T)
    logs = {}
    m.log = lambda key, val, **k: logs.setdefault(key, []).append(float(val))
    orig = torch.randn
    torch.randn = lambda *a, **k: torch.zeros(*a, **{kk: vv for kk, vv in k.items() if kk in ("device", "dtype")})
    torch.manual_seed(6)
    out = m.forward(fake_batch_kitti(seed=4), "train")
    torch.randn = orig
    blob = {"total_loss": np.float64(out["total_loss"].item())}
    for k, v in logs.items():
        blob["log/" + k] = np.asarray(v, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "kitti_forward.npz"), **blob)
    print({k: v for k, v in blob.items()})
