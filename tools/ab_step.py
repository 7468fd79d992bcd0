"""A/B of a module-level knob on ONE GPU box inside ONE process: the bench's training step (channels-last maps, device RNG) timed in
alternating blocks A B A B ... so that clock / thermal drift and box-to-box spread (+-4 %) cancel.
usage: ab_step.py <module.attr>=<v0>,<v1>[,...] [rounds] [steps]      e.g.  ab_step.py renderer.DEFER_HEAD_PACK=0,1 4 40
                                                                            ab_step.py "cfg.bwd_kernel='wide','wide_staged'" 4 40
A value is parsed with ast.literal_eval; `module` is relative to scenerf_amd."""
import argparse, ast, importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from scenerf_amd import synth

spec = sys.argv[1]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
path, vals = spec.split("=")
modname, attr = path.rsplit(".", 1)
mod = None if modname == "cfg" else importlib.import_module("scenerf_amd." + modname)     # cfg.<field>: the model's RenderConfig
vals = [ast.literal_eval(v) for v in vals.split(",")]

dev = torch.device("cuda:0")
args = argparse.Namespace(samples=128, precision="bf16", host_rng=False, optimizer=os.environ.get("AB_OPT", "fused"),
                          loss=os.environ.get("AB_LOSS", "source"))
R = 1200
torch.manual_seed(1)
model = bench.make_model(args, dev)
params = list(model.mlp.parameters()) + list(model.mlp_gaussian.parameters())
opt = bench.make_optimizer(args, params)
maps = bench._make_maps("hwc", dev, 0)
K, T = synth.kitti_cam_K().to(dev), synth.rel_pose(1.0, 0.0).to(dev)
pix = synth.stride2_pixels((1220, 370), R, 100).to(dev)
loss_fn = bench.make_loss(args, dev, (1220, 370), K, pix)


def step():
    for v in maps.values():
        v.grad = None
    out = model.render_rays_batch(K, T, maps, T_cam2velo=None, sampled_pixels=pix, ray_batch_size=R)
    loss = loss_fn(out)
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
    return loss


res = {repr(v): [] for v in vals}
for r in range(rounds):
    for v in vals:
        setattr(mod if mod is not None else model.render_cfg, attr, v)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        res[repr(v)].append((time.perf_counter() - t0) / steps * 1e3)
for k, ts in res.items():
    print("%s = %-8s  mean %.3f ms/step   rounds: %s" % (path, k, sum(ts) / len(ts), " ".join("%.3f" % t for t in ts)), flush=True)
