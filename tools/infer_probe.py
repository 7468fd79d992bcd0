"""Novel-view inference throughput (no_grad) through SceneRF.render_rays_batch: rays/s at a few (N, chunk) settings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scenerf_amd import synth
from scenerf_amd.model import SceneRF

dev = "cuda"
maps = {k: v.to(dev) for k, v in synth.feature_maps(1500, 452, 3).items()}
K, T = synth.kitti_cam_K().to(dev), synth.rel_pose(2.0, 10.0).to(dev)
for (U, P, nrays, chunk) in [(32, 8, 50468, 4000), (32, 8, 112850, 8192), (64, 16, 112850, 8192), (256, 64, 32768, 4096)]:
    m = SceneRF(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8, n_pts_uni=U, n_pts_per_gaussian=P, precision="bf16",
                device_rng=True).to(dev).eval()
    m.mlp.load_state_dict(synth.mlp_state(1, 4)); m.mlp_gaussian.load_state_dict(synth.mlp_state(2, 2, out_scale=4.0))
    pix = synth.stride2_pixels((1220, 370), min(nrays, 112850), 7).to(dev)
    if nrays > pix.shape[0]:
        pix = pix.repeat((nrays + pix.shape[0] - 1) // pix.shape[0], 1)[:nrays]
    with torch.no_grad():
        m.render_rays_batch(K, T, maps, sampled_pixels=pix[:chunk], ray_batch_size=chunk)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = m.render_rays_batch(K, T, maps, sampled_pixels=pix, ray_batch_size=chunk)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("N=%d rays=%d chunk=%d: %.1f ms  %.0f rays/s  (%.1f Msamples/s) peak mem %.1f GB" % (
        U + 4 * P, nrays, chunk, dt * 1e3, nrays / dt, nrays * (U + 4 * P) / dt / 1e6, torch.cuda.max_memory_allocated() / 2**30))
