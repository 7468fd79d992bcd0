"""Deterministic synthetic RGB-D scene for the TSDF tests: a tilted wall and a sphere seen from three poses (numpy PCG64)."""
import numpy as np


def make(seed=7, im_h=60, im_w=80, n_frames=3):
    rng = np.random.Generator(np.random.PCG64(seed))
    fx = fy = 70.0
    K = np.array([[fx, 0, im_w / 2 - 0.5], [0, fy, im_h / 2 - 0.5], [0, 0, 1]], dtype=np.float64)
    frames = []
    v, u = np.meshgrid(np.arange(im_h), np.arange(im_w), indexing="ij")
    rays = np.stack([(u - K[0, 2]) / fx, (v - K[1, 2]) / fy, np.ones_like(u, dtype=np.float64)], axis=-1)
    for i in range(n_frames):
        ang = np.deg2rad(-8.0 + 8.0 * i)
        R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
        t = np.array([0.15 * i - 0.1, 0.02 * i, 0.05 * i])
        pose = np.eye(4)
        pose[:3, :3], pose[:3, 3] = R, t
        # analytic depth: wall z_w = 3 + 0.2 x_w, sphere centre (0.2, 0.0, 2.2) radius 0.5, in world coordinates
        d = (R @ rays.reshape(-1, 3).T).T            # ray directions in world
        o = t
        tw = (3.0 + 0.2 * o[0] - o[2]) / (d[:, 2] - 0.2 * d[:, 0])
        c = np.array([0.2, 0.0, 2.2]); oc = o - c
        b = (d * oc).sum(1); a = (d * d).sum(1); cc = (oc * oc).sum() - 0.25
        disc = b * b - a * cc
        ts = np.where(disc > 0, (-b - np.sqrt(np.maximum(disc, 0))) / a, np.inf)
        tt = np.minimum(np.where(tw > 0, tw, np.inf), np.where(ts > 0, ts, np.inf))
        depth = np.where(np.isfinite(tt), tt, 0.0).reshape(im_h, im_w)      # z-depth because rays have z = 1 in the camera frame
        depth[rng.random((im_h, im_w)) < 0.03] = 0.0                          # invalid pixels
        color = rng.integers(0, 256, size=(im_h, im_w, 3)).astype(np.uint8)
        frames.append(dict(color=color, depth=depth.astype(np.float32), pose=pose))
    vol_bnds = np.array([[-1.5, 1.5], [-1.0, 1.0], [0.5, 4.0]])
    return dict(cam_intr=K, frames=frames, vol_bnds=vol_bnds, voxel_size=0.08, trunc_margin=0.4)
