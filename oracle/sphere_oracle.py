"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the decoder's image->sphere resampling, SURVEY §8f-2:
``DecoderSphere.get_sphere_feature`` (reference scenerf/models/unet2d_sphere.py:138-165).  Only tests/, tools/ probes and
__graft_entry__.smoke() may import this; the product (scenerf_amd.sphere -> csrc/sphere.hip) never does.

Pinned: tests/golden/sphere_resample.npz holds outputs and input gradients of the reference's own method run here on the CPU
with one thread (tests/golden/make_golden_sphere.py); tests/test_sphere.py checks this restatement against them.

What the reference does, per call (6 calls per image, one per encoder level):
  1. a (out_W, out_H, 2) map initialised to -10 receives ``pix // scale`` at the cells ``round(pix_sphere / scale)`` (clamped)
     -- an index_put with duplicates: on one CPU thread the LAST pixel in index order wins (on a GPU the winner is undefined);
     this restatement and the HIP path fix "last wins";
  2. the map is normalised to [-1, 1] and fed to ``F.grid_sample(bilinear, zeros padding, align_corners=False)``: since the map
     holds integer pixel coordinates the sample point is (sx - 0.5, sy - 0.5) up to fp32 rounding, i.e. (nearly) the mean of the
     2x2 block whose lower-right pixel is (sx, sy); empty cells (-10) fall outside and give 0;
  3. the result is viewed as (B, C, out_H, out_W).
"""
from __future__ import annotations

import numpy as np

EMPTY = -1


def scaled_dims(out_img_W: int, out_img_H: int, scale: int):
    """unet2d_sphere.py:139 -- Python's round (half to even): 1500x452 at scale 8 is 188x56."""
    return round(out_img_W / scale), round(out_img_H / scale)


def build_map(pix: np.ndarray, pix_sphere: np.ndarray, scale: int, out_w: int, out_h: int) -> np.ndarray:
    """unet2d_sphere.py:140-147.  pix (P,2) float32 pixel coordinates, pix_sphere (P,2) int64 sphere coordinates.
    Returns src (out_h, out_w) int32: (sy << 16) | sx of the winning pixel's ``pix // scale`` or -1 for an empty cell."""
    pix = np.asarray(pix, dtype=np.float32)
    ps = np.asarray(pix_sphere, dtype=np.int64)
    s = np.float32(scale)
    u = np.rint(ps[:, 0].astype(np.float32) / s).astype(np.int64).clip(0, out_w - 1)    # torch.round: half to even
    v = np.rint(ps[:, 1].astype(np.float32) / s).astype(np.int64).clip(0, out_h - 1)
    sx = np.floor(pix[:, 0] / s).astype(np.int64)
    sy = np.floor(pix[:, 1] / s).astype(np.int64)
    winner = np.full(out_h * out_w, -1, dtype=np.int64)
    np.maximum.at(winner, v * out_w + u, np.arange(pix.shape[0], dtype=np.int64))      # last index wins
    src = np.full(out_h * out_w, EMPTY, dtype=np.int32)
    m = winner >= 0
    src[m] = ((sy[winner[m]] << 16) | sx[winner[m]]).astype(np.int32)
    return src.reshape(out_h, out_w)


def _sample_point(s_int: np.ndarray, size: int):
    """map value -> grid_sample pixel coordinate, op by op in fp32: unet2d_sphere.py:151-153 then ATen's
    grid_sampler_unnormalize (align_corners=False): ((g + 1) * size - 1) / 2."""
    f = np.float32
    g = s_int.astype(f) / f(size)
    g = g * f(2)
    g = g - f(1)
    t = g + f(1)
    t = t * f(size)
    t = t - f(1)
    return t / f(2)


def taps(src: np.ndarray, H: int, W: int):
    """Per cell: the four tap pixel indices (iy*W+ix, or -1 outside / empty) and their fp32 weights, order nw, ne, sw, se."""
    src = src.reshape(-1)
    valid = src >= 0
    sx = np.where(valid, src & 0xFFFF, 0)
    sy = np.where(valid, src >> 16, 0)
    ix, iy = _sample_point(sx, W), _sample_point(sy, H)
    x0, y0 = np.floor(ix), np.floor(iy)
    x1, y1 = x0 + 1, y0 + 1
    w = np.stack([(x1 - ix) * (y1 - iy), (ix - x0) * (y1 - iy), (x1 - ix) * (iy - y0), (ix - x0) * (iy - y0)], 0).astype(np.float32)
    xs = np.stack([x0, x1, x0, x1], 0).astype(np.int64)
    ys = np.stack([y0, y0, y1, y1], 0).astype(np.int64)
    inside = (xs >= 0) & (xs < W) & (ys >= 0) & (ys < H) & valid[None]
    idx = np.where(inside, ys * W + xs, -1)
    return idx, w


def resample_forward(x: np.ndarray, src: np.ndarray) -> np.ndarray:
    """unet2d_sphere.py:149-165.  x (B,C,H,W) float32 -> (B,C,out_h,out_w)."""
    B, C, H, W = x.shape
    out_h, out_w = src.shape
    idx, w = taps(src, H, W)
    flat = x.reshape(B * C, H * W).astype(np.float32)
    out = np.zeros((B * C, out_h * out_w), dtype=np.float32)
    for t in range(4):
        val = np.where(idx[t][None] >= 0, flat[:, np.maximum(idx[t], 0)], np.float32(0))
        out = out + val * w[t][None]
    return out.reshape(B, C, out_h, out_w)


def resample_backward(dout: np.ndarray, src: np.ndarray, H: int, W: int) -> np.ndarray:
    """Adjoint of resample_forward w.r.t. x (the map carries no gradient), accumulated in float64."""
    B, C, out_h, out_w = dout.shape
    idx, w = taps(src, H, W)
    d = dout.reshape(B * C, -1).astype(np.float64)
    dx = np.zeros((B * C, H * W), dtype=np.float64)
    for t in range(4):
        m = idx[t] >= 0
        np.add.at(dx, (slice(None), idx[t][m]), d[:, m] * w[t][m].astype(np.float64)[None])
    return dx.reshape(B, C, H, W)


def csr_of(src: np.ndarray, H: int, W: int):
    """Cells grouped by their source pixel, ascending cell order inside a group: the deterministic gather form of the backward
    pass used by the HIP kernel.  Groups are indexed sy*(W+1)+sx on a (H+1) x (W+1) grid (an entry one past the plane still has
    in-range taps; further out it has none and is dropped).  Returns (row_ptr [(H+1)*(W+1)+1] int32, cells [n] int32)."""
    flat = src.reshape(-1)
    cells = np.nonzero(flat >= 0)[0]
    sy, sx = (flat[cells] >> 16).astype(np.int64), (flat[cells] & 0xFFFF).astype(np.int64)
    keep = (sy <= H) & (sx <= W)
    cells, q = cells[keep], (sy * (W + 1) + sx)[keep]
    order = np.argsort(q, kind="stable")
    n = (H + 1) * (W + 1)
    row_ptr = np.zeros(n + 1, dtype=np.int32)
    row_ptr[1:] = np.cumsum(np.bincount(q, minlength=n))
    return row_ptr, cells[order].astype(np.int32)


def resample_backward_gather(dout: np.ndarray, row_ptr: np.ndarray, cells: np.ndarray, H: int, W: int) -> np.ndarray:
    """The backward pass in the HIP kernel's own form and summation order (fp32): every source pixel (px, py) sums, for the four
    cell groups (px+ex, py+ey), ey-major, the cells of the group in ascending order, dout * weight.  Equals the adjoint above up to
    fp32 rounding; exists so the GPU test can demand bit equality."""
    f = np.float32
    B, C, out_h, out_w = dout.shape
    d = dout.reshape(B * C, -1).astype(f)
    py, px = np.divmod(np.arange(H * W, dtype=np.int64), W)
    acc = np.zeros((B * C, H * W), dtype=f)
    for ey in (0, 1):
        for ex in (0, 1):
            qx, qy = px + ex, py + ey
            q = qy * (W + 1) + qx
            e0, e1 = row_ptr[q].astype(np.int64), row_ptr[q + 1].astype(np.int64)
            ix, iy = _sample_point(qx, W), _sample_point(qy, H)
            x0, y0 = np.floor(ix), np.floor(iy)
            tx, ty = px - x0.astype(np.int64), py - y0.astype(np.int64)
            ok = (tx >= 0) & (tx <= 1) & (ty >= 0) & (ty <= 1)
            wx = np.where(tx == 1, ix - x0, x0 + f(1) - ix).astype(f)
            wy = np.where(ty == 1, iy - y0, y0 + f(1) - iy).astype(f)
            w = (wx * wy).astype(f)
            r = 0
            while True:
                m = ok & (e0 + r < e1)
                if not m.any():
                    break
                cell = cells[(e0 + r)[m]]
                acc[:, m] = acc[:, m] + d[:, cell] * w[m][None]
                r += 1
    return acc.reshape(B, C, H, W)
