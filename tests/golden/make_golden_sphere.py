"""Golden vectors for the decoder's image->sphere resampling from the reference itself:
``DecoderSphere.get_sphere_feature`` (scenerf/models/unet2d_sphere.py:138-165) called unbound on the CPU with ONE thread (its
index_put with duplicate cells is then sequential: the last pixel wins).  Build container only.

  small/*   a 1/10-size KITTI geometry (122x37 image, 150x45 sphere; pix is the row-major (x, y) pixel grid): inputs, outputs and input gradients for scales 1, 2, 4, 8;
  kitti/*   the full KITTI geometry (1220x370 -> 1500x452): the reference's own scattered map for every level the decoder uses
            (1 ... 32), captured from the grid it hands to F.grid_sample and stored as delta-coded int32 ``src`` maps, plus the
            sphere coordinates it was built from (delta-coded int16).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _install_reference, REF   # noqa: E402

KITTI_K = np.array([[707.0912, 0.0, 601.8873], [0.0, 707.0912, 183.1104], [0.0, 0.0, 1.0]], dtype=np.float32)
ANGLES = dict(v_angle_max=104.7294 + 8, v_angle_min=75.4815 - 8, h_angle_max=131.1128 + 20, h_angle_min=49.5950 - 20)


def plane_dims(img_W, img_H, scale):
    """EfficientNet 'same' padding: every stride-2 stage rounds up."""
    w, h = img_W, img_H
    s = 1
    while s < scale:
        w, h, s = (w + 1) // 2, (h + 1) // 2, s * 2
    return w, h


if __name__ == "__main__":
    assert os.path.isdir(REF)
    _install_reference()
    torch.set_num_threads(1)
    import torch.nn.functional as F
    from scenerf.models.spherical_mapping import SphericalMapping
    from scenerf.models.unet2d_sphere import DecoderSphere
    import scenerf.models.unet2d_sphere as U
    blob = {}

    # ---- small geometry: full data
    img_W, img_H, out_W, out_H = 122, 37, 150, 45
    K = KITTI_K.copy(); K[:2] /= 10
    sm = SphericalMapping(img_W=img_W, img_H=img_H, out_img_W=out_W, out_img_H=out_H, **ANGLES)
    pix, pix_sphere, _ = sm.from_pixels(inv_K=torch.inverse(torch.from_numpy(K)))
    blob["small/pix_sphere"] = pix_sphere.numpy().astype(np.int16)
    blob["small/dims"] = np.array([img_W, img_H, out_W, out_H], dtype=np.int32)
    me = types.SimpleNamespace(out_img_W=out_W, out_img_H=out_H)
    rng = np.random.Generator(np.random.PCG64(21))
    for scale, C in ((1, 2), (2, 5), (4, 4), (8, 6)):
        w, h = plane_dims(img_W, img_H, scale)
        x = torch.from_numpy(rng.standard_normal((1, C, h, w), dtype=np.float32)).requires_grad_(True)
        out = DecoderSphere.get_sphere_feature(me, x, pix, pix_sphere, scale)
        g = torch.from_numpy(rng.standard_normal(tuple(out.shape), dtype=np.float32))
        (out * g).sum().backward()
        blob[f"small/s{scale}/x"] = x.detach().numpy()
        blob[f"small/s{scale}/out"] = out.detach().contiguous().numpy()
        blob[f"small/s{scale}/g"] = g.numpy()
        blob[f"small/s{scale}/dx"] = x.grad.numpy()

    # ---- KITTI geometry: the reference's own map per level
    img_W, img_H, out_W, out_H = 1220, 370, 1500, 452
    sm = SphericalMapping(img_W=img_W, img_H=img_H, out_img_W=out_W, out_img_H=out_H, **ANGLES)
    pix, pix_sphere, _ = sm.from_pixels(inv_K=torch.inverse(torch.from_numpy(KITTI_K)))
    ps = pix_sphere.numpy().astype(np.int16).reshape(img_H, img_W, 2)
    blob["kitti/pix_sphere_d"] = np.diff(ps, axis=1, prepend=0).astype(np.int16)      # delta along a row: compresses to almost nothing
    me = types.SimpleNamespace(out_img_W=out_W, out_img_H=out_H)
    captured = {}
    real = F.grid_sample

    def spy(x, grid, **kw):
        captured["grid"] = grid.detach().clone()
        return real(x, grid, **kw)
    U.F.grid_sample = spy
    for scale in (1, 2, 4, 8, 16, 32):
        w, h = plane_dims(img_W, img_H, scale)
        x = torch.ones(1, 1, h, w)
        out = DecoderSphere.get_sphere_feature(me, x, pix, pix_sphere, scale)
        ow, oh = round(out_W / scale), round(out_H / scale)
        assert tuple(out.shape) == (1, 1, oh, ow)
        g = captured["grid"].reshape(ow, oh, 2).double()
        mx = torch.round((g[..., 0] + 1) / 2 * w).long()
        my = torch.round((g[..., 1] + 1) / 2 * h).long()
        src = torch.where(mx < 0, torch.full_like(mx, -1), (my << 16) | mx).T.contiguous().numpy().astype(np.int32)   # (oh, ow)
        blob[f"kitti/s{scale}/src_d"] = np.diff(src, axis=1, prepend=0).astype(np.int32)
        blob[f"kitti/s{scale}/plane"] = np.array([w, h], dtype=np.int32)
        blob[f"kitti/s{scale}/ones_x4"] = np.rint(out[0, 0].numpy() * 4).astype(np.uint8)
    U.F.grid_sample = real
    path = os.path.join(HERE, "sphere_resample.npz")
    np.savez_compressed(path, **blob)
    print(os.path.getsize(path), {k: np.shape(v) for k, v in blob.items()})
