"""Where does the HOST time of one eagerly issued training step go?  cProfile over N eager steps of bench.py's own step (the driver's
configuration), GPU work queued asynchronously -- the profile is the issue cost.   usage: host_profile.py [steps]"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from scenerf_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
sys.argv = ["bench.py"]
args = bench.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
model = bench.make_model(args, dev)
params = list(model.mlp.parameters()) + list(model.mlp_gaussian.parameters())
opt = bench.make_optimizer(args, params)
maps = bench._make_maps(args.maps, dev, 0)
K, T = synth.kitti_cam_K().to(dev), synth.rel_pose(1.0, 0.0).to(dev)
R = args.rays
pix = synth.stride2_pixels((1220, 370), R, 100).to(dev)
loss_fn = bench.make_loss(args, dev, (1220, 370), K, pix, 0)


def step():
    for v in maps.values():
        v.grad = None
    out = model.render_rays_batch(K, T, maps, T_cam2velo=None, sampled_pixels=pix, ray_batch_size=R)
    loss = loss_fn(out)
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
    return loss


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("eager step: host issue %.3f ms, wall %.3f ms per step (%d steps)" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, n))
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step()
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    print("\n".join(l[:170] for l in s.getvalue().splitlines()[:45]))
