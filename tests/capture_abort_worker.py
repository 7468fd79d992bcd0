"""Worker of tests/test_gpu_graph.py::test_a_failed_capture_leaves_a_process_that_can_step_eagerly (its own process: a recovery that fell
short would poison every later test of the suite)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_gpu_graph as tg
from scenerf_amd.graph import GraphedStep, build_on_all_ranks

m, opt, maps, K, T, pix, noise = tg._setup(17)
calls = []


def bad_loss(out):
    calls.append(1)
    if len(calls) > 1:                      # (the warm-up step passes; the captured one syncs)
        float(out["depth"].detach().mean())   # device -> host inside the capture: not capturable
    return tg._loss(out)


cur = torch.cuda.current_stream()
g, note = build_on_all_ranks(lambda: GraphedStep(m, opt, bad_loss, K, T, maps, pix, ray_batch_size=256, warmup=1, noise=noise))
ok_state = g is None and "capture failed on this rank" in note and torch.cuda.current_stream() == cur and not torch.cuda.is_current_stream_capturing()
losses = []
for _ in range(2):                          # ... and the process still renders, differentiates and steps
    opt.zero_grad(set_to_none=True)
    for v in maps.values():
        v.grad = None
    loss = tg._loss(m.render_rays_batch(K, T, maps, T_cam2velo=None, sampled_pixels=pix, ray_batch_size=256, noise=noise))
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    losses.append(float(loss.detach()))
print("CAPTURE_ABORT state_ok=%s finite=%s note=%r" % (ok_state, all(torch.isfinite(torch.tensor(losses))), note[:80]))
