"""Golden values for the BundleFusion training forward (loss assembly, scenerf/models/scenerf_bf.py:124-247) from the reference
itself: its ``SceneRFBundleFusion.forward`` is run on the CPU with the renderer and the encoder replaced by the deterministic fakes
of ``bf_fakes.py`` (the renderer is GPU-only here and is pinned elsewhere; this pins the CALLER: ray sampling, losses, weights,
depth metrics).  The identity-reprojection noise (torch.randn * 1e-5) is replaced by zeros.  Build container only."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _install_reference, REF   # noqa: E402
from bf_fakes import FakeNetRgb, fake_batch, fake_render   # noqa: E402

if __name__ == "__main__":
    assert os.path.isdir(REF)
    _install_reference()
    from scenerf.models.scenerf_bf import SceneRF as SceneRFBundleFusion   # (the indoor model is also called SceneRF in its own file)
    kw = dict(som_sigma=2.0, img_size=(64, 48), n_rays=256, sample_grid_size=2, sphere_H=48, sphere_W=64, eval_depth=10)
    m = SceneRFBundleFusion(**kw)
    m.net_rgb = FakeNetRgb()
    m.render_rays_batch = lambda cam_K, T, x_rgb, ray_batch_size=None, sampled_pixels=None, **k: fake_render(sampled_pixels, T)
    logs = {}
    m.log = lambda key, val, **k: logs.setdefault(key, []).append(float(val))
    orig = torch.randn
    torch.randn = lambda *a, **k: torch.zeros(*a, **{kk: vv for kk, vv in k.items() if kk in ("device", "dtype")})
    torch.manual_seed(5)
    out = m.forward(fake_batch(seed=3), "train")
    torch.randn = orig
    blob = {"total_loss": np.float64(out["total_loss"].item())}
    for k, v in logs.items():
        blob["log/" + k] = np.asarray(v, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "bf_forward.npz"), **blob)
    print({k: v for k, v in blob.items()})
