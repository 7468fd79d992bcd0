"""Do a replayed step's gradients equal an eagerly issued step's?  Same start, same static noise and pixels, no optimizer: per-tensor relative
L2 of the parameter gradients and map gradients (eager vs eager = the atomics' run-to-run level; replay vs eager must sit on it)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_graph as tg
from scenerf_amd.graph import GraphedStep

R = int(sys.argv[1]) if len(sys.argv) > 1 else 256


def setup():
    m, opt, maps, K, T, pix, noise = tg._setup(21)
    return m, maps, K, T, pix, noise


def grads_eager():
    m, maps, K, T, pix, noise = setup()
    tg._loss(m.render_rays_batch(K, T, maps, T_cam2velo=None, sampled_pixels=pix, ray_batch_size=256, noise=noise)).backward()
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}, {k: v.grad.clone() for k, v in maps.items()}


def grads_replay():
    m, maps, K, T, pix, noise = setup()
    gs = GraphedStep(m, None, tg._loss, K, T, maps, pix, ray_batch_size=256, warmup=1, noise=noise)
    gs(); torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}, {k: v.clone() for k, v in gs.map_grads.items()}


a, am = grads_eager()
b, bm = grads_eager()
c, cm = grads_replay()
worst = {}
for nm in a:
    den = float(a[nm].norm()) + 1e-30
    worst[nm] = (float((a[nm] - b[nm]).norm()) / den, float((a[nm] - c[nm]).norm()) / den)
for nm in am:
    den = float(am[nm].norm()) + 1e-30
    worst["x_rgb." + nm] = (float((am[nm] - bm[nm]).norm()) / den, float((am[nm] - cm[nm]).norm()) / den)
top = sorted(worst.items(), key=lambda kv: -kv[1][1])[:8]
print("relative L2 of gradients, (eager vs eager, replay vs eager), worst tensors:")
for nm, (e, r) in top:
    print("  %-40s %.2e  %.2e" % (nm, e, r))
