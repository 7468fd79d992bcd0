"""Does a whole training step (render_rays_batch forward + backward into both MLPs and the maps + fused AdamW) capture into ONE hipGraph?
Captures the bench step with device RNG and a capturable optimizer, replays it, compares the replayed loss trajectory with eager steps
from the same start, and times both.   usage: graph_step_probe.py [steps]"""
import argparse, copy, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from scenerf_amd import synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
args = argparse.Namespace(samples=128, precision="bf16")
R = 1200


def setup():
    torch.manual_seed(1)
    model = bench.make_model(args, dev)
    model.render_cfg.device_rng = True          # no host-side normal draw inside the captured region
    params = list(model.mlp.parameters()) + list(model.mlp_gaussian.parameters())
    opt = torch.optim.AdamW(params, lr=1e-5, weight_decay=0.0, fused=True, capturable=True)
    maps = bench._make_maps("hwc", dev, 0)
    K, T = synth.kitti_cam_K().to(dev), synth.rel_pose(1.0, 0.0).to(dev)
    pix = synth.stride2_pixels((1220, 370), R, 100).to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        for v in maps.values():
            v.grad = None
        out = model.render_rays_batch(K, T, maps, T_cam2velo=None, sampled_pixels=pix, ray_batch_size=R)
        loss = out["depth"].mean() + out["color"].mean() + out["loss_kl"].mean() + out["gaussian_means"].mean()
        loss.backward()
        opt.step()
        return loss
    return model, step


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


model, step = setup()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):      # warm-up on a side stream (allocator, first-use setup), as torch's capture recipe asks
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
eager_ms = timed(step, steps)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        static_loss = step()
except Exception as e:
    print("CAPTURE FAILED: %s: %s" % (type(e).__name__, str(e)[:600]))
    sys.exit(0)
g.replay()
torch.cuda.synchronize()
print("captured: loss after the first replay %.6f (finite: %s)" % (float(static_loss), bool(torch.isfinite(static_loss))))
graph_ms = timed(g.replay, steps)
print("whole training step, R = %d: eager %.3f ms/step, one hipGraph replay %.3f ms/step" % (R, eager_ms, graph_ms))
