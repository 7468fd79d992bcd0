"""Scale-activity masks of the row tiles in one bench step (which levels each 128-row tile touches), as the feature-gradient kernel sees them."""
import argparse, collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from scenerf_amd import synth
dev = torch.device("cuda:0")
args = argparse.Namespace(samples=128, precision="bf16")
model = bench.make_model(args, dev)
model.debug_aux = True
maps = {k: v.to(dev) for k, v in synth.feature_maps(1500, 452, 3).items()}
K, T = synth.kitti_cam_K().to(dev), synth.rel_pose(1.0, 0.0).to(dev)
pix = synth.stride2_pixels((1220, 370), 1200, 100).to(dev)
with torch.no_grad():
    model.render_rays_batch(K, T, maps, T_cam2velo=None, sampled_pixels=pix, ray_batch_size=1200)
tm = model.last_aux["tile_mask"].cpu()[:1200].tolist()
h = collections.Counter(tm)
print("tile masks (value: count):", dict(sorted(h.items())))
act = [i for i, m in enumerate(tm) if m & 30]
print("tiles with a coarser level: %d of %d; per XCD slot (tile %% 8): %s" % (len(act), len(tm), [sum(1 for i in act if i % 8 == x) for x in range(8)]))
runs = []
cur = 0
for m in tm:
    if m & 30:
        cur += 1
    elif cur:
        runs.append(cur); cur = 0
if cur: runs.append(cur)
print("runs of consecutive coarser-level tiles: n=%d, longest %d" % (len(runs), max(runs) if runs else 0))
