"""Deterministic stand-ins for the two GPU/hub-bound pieces around the BundleFusion loss assembly (used by make_golden_bf_forward.py
with the reference and by tests/test_loss_side.py with scenerf_amd): a renderer that is a closed-form function of the sampled
pixels, an encoder that returns empty maps, and a synthetic batch in the BundleFusion collate layout."""
import numpy as np
import torch


def fake_render(pix, T):
    """What render_rays_batch returns, as smooth functions of the pixel coordinates (R = pix.shape[0], G = 4 gaussians)."""
    x, y = pix[:, 0].float() / 64.0, pix[:, 1].float() / 48.0
    depth = 1.5 + 2.0 * x + 1.0 * y + 0.1 * T[2, 3]
    g = torch.arange(4, dtype=torch.float32)[None, :]
    means = depth[:, None] * (0.6 + 0.25 * g) + 0.05 * torch.sin(7 * x)[:, None]
    return dict(depth=depth, color=torch.stack([x, y, 0.5 * (x + y)], 1), loss_kl=0.1 + 0.2 * x * y,
                gaussian_means=means, gaussian_stds=0.3 + 0.1 * g + 0.0 * means, som_vars=0.5 + 0.05 * g * x[:, None],
                weights_at_depth=0.2 + 0.5 * y, closest_pts_to_depths=0.05 + 0.1 * x)


class FakeNetRgb(torch.nn.Module):
    def forward(self, img, pix=None, pix_sphere=None):
        return {"1_1": torch.zeros(img.shape[0], 1, 2, 2)}


def fake_batch(seed, bs=2, n_src=2, H=48, W=64):
    rng = np.random.Generator(np.random.PCG64(seed))
    K = torch.tensor([[60.0, 0, 32.0], [0, 60.0, 24.0], [0, 0, 1]])

    def pose(tz):
        T = torch.eye(4)
        T[0, 3], T[2, 3] = 0.05, tz
        return T
    depth = lambda: np.where(rng.random((H, W)) < 0.8, rng.uniform(0.5, 6.0, (H, W)), 0.0).astype(np.float32)   # (numpy, as the dataset hands it over)
    img = lambda: torch.from_numpy(rng.random((3, H, W), dtype=np.float32))
    return dict(img_inputs=torch.zeros(bs, 3, H, W), cam_K_depth=[K for _ in range(bs)],
                T_source2targets=[[pose(0.1 * (s + 1)) for s in range(n_src)] for _ in range(bs)],
                T_source2infers=[[pose(0.3 * (s + 1)) for s in range(n_src)] for _ in range(bs)],
                img_sources=[[img() for _ in range(n_src)] for _ in range(bs)],
                img_targets=[[img() for _ in range(n_src)] for _ in range(bs)],
                source_depths=[[depth() for _ in range(n_src)] for _ in range(bs)])


def fake_batch_kitti(seed, bs=2, n_src=2, H=48, W=64, n_lidar=40):
    """The KITTI collate layout (scenerf.py:119-201): per-sample intrinsics, velodyne extrinsics, lidar pixels + depths."""
    b = fake_batch(seed, bs=bs, n_src=n_src, H=H, W=W)
    rng = np.random.Generator(np.random.PCG64(seed + 100))
    b["cam_K"] = b.pop("cam_K_depth")
    b.pop("source_depths")
    Tv = torch.eye(4)
    Tv[:3, 3] = torch.tensor([0.1, -0.2, 0.3])
    b["T_velo_2_cam"] = [Tv for _ in range(bs)]
    b["loc2d_with_depths"] = [[torch.from_numpy(np.stack([rng.integers(0, W, n_lidar), rng.integers(0, H, n_lidar)], 1).astype(np.int64))
                               for _ in range(n_src)] for _ in range(bs)]
    b["lidar_depths"] = [[torch.from_numpy(rng.uniform(1.0, 9.0, n_lidar).astype(np.float32)) for _ in range(n_src)] for _ in range(bs)]
    return b
