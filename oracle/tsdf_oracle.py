"""CPU restatement of the reference's TSDF fusion (TEST INFRASTRUCTURE ONLY -- never imported by the product path; see oracle/README.md).

reference: scenerf/data/utils/fusion.py.  The file holds two different update rules:
  * ``semantics="gpu"``: the pycuda kernel (fusion.py:72-145) -- what runs when pycuda imports (``use_gpu=True`` is the default of
    every reconstruction script).  The kernel text is plain CUDA C: oracle/build_ref.py compiles it verbatim with hipcc (every operation
    rounded on its own) and tests/golden/make_golden_tsdf_gpu.py runs it on an MI355X; this restatement follows the text line by line
    in float32 and is **pinned** bit for bit on those volumes (tests/golden/tsdf_gpu_semantics.npz, tests/test_tsdf.py).
  * ``semantics="cpu"``: the vectorised CPU path (fusion.py:236-325 with the helpers :152-203).  **Pinned** against the reference itself
    (tests/golden/make_golden_tsdf.py imports fusion.py with numba / skimage stubs and runs this path; tests/test_tsdf.py).
Volumes are float32 [X][Y][Z]; colours are folded as floor(b*65536 + g*256 + r).
"""
import numpy as np

COLOR_CONST = 256 * 256


def new_volume(vol_bnds, voxel_size):
    """fusion.py:33-56: dims = ceil(extent / voxel), origin = lower bounds (float32), tsdf = 255, weight = colour = 0."""
    vol_bnds = np.asarray(vol_bnds, dtype=np.float64).copy()
    dim = np.ceil((vol_bnds[:, 1] - vol_bnds[:, 0]) / float(voxel_size)).astype(int)
    origin = vol_bnds[:, 0].astype(np.float32)
    tsdf = np.zeros(dim, dtype=np.float32) + 255
    return dim, origin, tsdf, np.zeros(dim, dtype=np.float32), np.zeros(dim, dtype=np.float32)


def fold_color(color_im):
    """fusion.py:218-220."""
    c = np.asarray(color_im).astype(np.float32)
    return np.floor(c[..., 2] * COLOR_CONST + c[..., 1] * 256 + c[..., 0]).astype(np.float32)


def _coords(dim):
    xv, yv, zv = np.meshgrid(range(dim[0]), range(dim[1]), range(dim[2]), indexing="ij")
    return np.stack([xv.reshape(-1), yv.reshape(-1), zv.reshape(-1)], axis=1)


def integrate_gpu_semantics(tsdf, weight, color, origin, voxel_size, color_folded, depth_im, cam_intr, cam_pose, trunc_margin,
                            obs_weight=1.0):
    """fusion.py:84-142, one 'thread' per voxel, float32 throughout; roundf = half away from zero."""
    f = np.float32
    dim = tsdf.shape
    vc = _coords(dim).astype(np.float32)
    K = np.asarray(cam_intr, dtype=np.float32).reshape(3, 3)
    P = np.asarray(cam_pose, dtype=np.float32).reshape(4, 4)
    vs = f(voxel_size)
    pt = origin.astype(np.float32)[None, :] + vc * vs                                # :96-99
    tmp = pt - P[:3, 3][None, :]                                                      # :101-103
    cam = np.stack([P[0, 0] * tmp[:, 0] + P[1, 0] * tmp[:, 1] + P[2, 0] * tmp[:, 2],
                    P[0, 1] * tmp[:, 0] + P[1, 1] * tmp[:, 1] + P[2, 1] * tmp[:, 2],
                    P[0, 2] * tmp[:, 0] + P[1, 2] * tmp[:, 1] + P[2, 2] * tmp[:, 2]], axis=1).astype(np.float32)   # :104-106

    def roundf(x):
        # C roundf: nearest integer, halves away from zero -- trunc(x) + (|x - trunc(x)| >= 0.5) with the sign of x.  (floor(x + 0.5)
        # is NOT it in float32: the sum itself rounds, e.g. 0.49999997 + 0.5 = 1.0)
        x = np.asarray(x, dtype=np.float32)
        t = np.trunc(x)
        return (t + np.copysign((np.abs(x - t) >= f(0.5)).astype(np.float32), x)).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        u = K[0, 0] * (cam[:, 0] / cam[:, 2]) + K[0, 2]
        v = K[1, 1] * (cam[:, 1] / cam[:, 2]) + K[1, 2]
    fin = np.isfinite(u) & np.isfinite(v)
    px = np.where(fin, roundf(np.where(fin, u, 0)), -1).astype(np.int64)              # :108-109
    py = np.where(fin, roundf(np.where(fin, v, 0)), -1).astype(np.int64)
    im_h, im_w = depth_im.shape
    ok = (px >= 0) & (px < im_w) & (py >= 0) & (py < im_h) & ~(cam[:, 2] < 0)        # :113-114
    depth = np.zeros(len(px), dtype=np.float32)
    depth[ok] = depth_im[py[ok], px[ok]].astype(np.float32)
    ok &= depth != 0                                                                  # :116-118
    diff = depth - cam[:, 2]
    ok &= ~(diff < -f(trunc_margin))                                                  # :121-123
    dist = np.minimum(f(1.0), diff / f(trunc_margin)).astype(np.float32)
    idx = np.nonzero(ok)[0]
    t, w, c = tsdf.reshape(-1), weight.reshape(-1), color.reshape(-1)
    w_old = w[idx]
    ow = f(obs_weight)
    w_new = (w_old + ow).astype(np.float32)
    w[idx] = w_new
    t[idx] = ((t[idx] * w_old + ow * dist[idx]) / w_new).astype(np.float32)          # :128-129
    old = c[idx]
    ob = np.floor(old / f(COLOR_CONST)); og = np.floor((old - ob * f(COLOR_CONST)) / f(256)); orr = old - ob * f(COLOR_CONST) - og * f(256)
    new = color_folded[py[idx], px[idx]].astype(np.float32)
    nb = np.floor(new / f(COLOR_CONST)); ng = np.floor((new - nb * f(COLOR_CONST)) / f(256)); nr = new - nb * f(COLOR_CONST) - ng * f(256)
    nb = np.minimum(roundf(((ob * w_old + ow * nb) / w_new).astype(np.float32)), f(255))   # :137-141
    ng = np.minimum(roundf(((og * w_old + ow * ng) / w_new).astype(np.float32)), f(255))
    nr = np.minimum(roundf(((orr * w_old + ow * nr) / w_new).astype(np.float32)), f(255))
    c[idx] = (nb * f(COLOR_CONST) + ng * f(256) + nr).astype(np.float32)


def integrate_cpu_semantics(tsdf, weight, color, origin, voxel_size, color_folded, depth_im, cam_intr, cam_pose, trunc_margin,
                            obs_weight=1.0):
    """fusion.py:236-325 with vox2world (:152-162), rigid_transform (:392-397), cam2pix (:166-185), integrate_tsdf (:188-203)."""
    dim = tsdf.shape
    vc = _coords(dim)
    # vox2world (:152-162): origin[j] + vox_size * coord on float32 arrays with a Python-float vox_size.  Under NumPy >= 2 (NEP 50) and
    # without numba's typing the product and the sum are float32 -- that is what produced the golden file; numba would form them in
    # float64 and round once (a last-bit difference in the world coordinate, i.e. ~1e-7 m in the stored distances)
    pts = origin.astype(np.float32)[None, :] + np.float32(voxel_size) * vc.astype(np.float32)
    T = np.linalg.inv(np.asarray(cam_pose))
    cam = np.dot(T, np.hstack([pts, np.ones((len(pts), 1), dtype=np.float32)]).T).T[:, :3]          # float64
    z = cam[:, 2]
    K = np.asarray(cam_intr).astype(np.float32)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        u = np.round(cam[:, 0] * fx / z + cx)
        v = np.round(cam[:, 1] * fy / z + cy)
    fin = np.isfinite(u) & np.isfinite(v)
    px = np.where(fin, u, -1).astype(np.int64)
    py = np.where(fin, v, -1).astype(np.int64)
    im_h, im_w = depth_im.shape
    valid_pix = (px >= 0) & (px < im_w) & (py >= 0) & (py < im_h) & (z > 0)                          # :247-251
    depth = np.zeros(px.shape)
    depth[valid_pix] = depth_im[py[valid_pix], px[valid_pix]]
    diff = depth - z
    valid = (depth > 0) & (diff >= -trunc_margin)                                                    # :256-258
    idx = np.nonzero(valid)[0]
    t, w, c = tsdf.reshape(-1), weight.reshape(-1), color.reshape(-1)
    keep = np.abs(t[idx].astype(np.float64)) < np.abs(diff[idx])                                     # :196-200
    w[idx] = (w[idx] + obs_weight).astype(np.float32)
    upd = idx[~keep]
    t[upd] = diff[upd].astype(np.float32)
    new = color_folded[py[upd], px[upd]]
    nb = np.floor(new / COLOR_CONST); ng = np.floor((new - nb * COLOR_CONST) / 256); nr = new - nb * COLOR_CONST - ng * 256
    c[upd] = (nb * COLOR_CONST + ng * 256 + nr).astype(np.float32)                                   # :283-298
