"""What state does a capture that fails HALF-WAY leave the process in, and what brings it back?  (ROCm 7.2: every later launch fails with
hipErrorStreamCaptureInvalidated / ...Implicit until the streams that were forked into the capture are gone.)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_graph as tg
from scenerf_amd import renderer
from scenerf_amd.graph import GraphedStep

m, opt, maps, K, T, pix, noise = tg._setup(17)
calls = []


def bad_loss(out):
    calls.append(1)
    if len(calls) > 1:
        float(out["depth"].mean())
    return tg._loss(out)


def eager_ok(tag):
    try:
        opt.zero_grad(set_to_none=True)
        for v in maps.values():
            v.grad = None
        loss = tg._loss(m.render_rays_batch(K, T, maps, T_cam2velo=None, sampled_pixels=pix, ray_batch_size=256, noise=noise))
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        print("%-40s eager step OK (loss %.4f)" % (tag, float(loss)))
        return True
    except Exception as e:      # noqa: BLE001
        print("%-40s eager step FAILS: %s" % (tag, str(e).splitlines()[0][:150]))
        return False


try:
    GraphedStep(m, opt, bad_loss, K, T, maps, pix, ray_batch_size=256, warmup=1, noise=noise)
    print("capture unexpectedly succeeded")
except Exception as e:          # noqa: BLE001
    print("capture failed as intended:", str(e).splitlines()[0][:120])
print("current stream capturing:", torch.cuda.is_current_stream_capturing())
if not eager_ok("as left by GraphedStep"):
    side = renderer._SIDE.copy()
    renderer._SIDE.clear()                       # new Python-side streams
    if not eager_ok("with fresh renderer side streams"):
        for s_ in side.values():
            try:
                print("  side stream capturing?", s_.query())
            except Exception as e:               # noqa: BLE001
                print("  side stream query:", str(e).splitlines()[0][:100])
        try:
            torch.cuda.synchronize()
        except Exception as e:                   # noqa: BLE001
            print("  device synchronize:", str(e).splitlines()[0][:100])
        eager_ok("after a device synchronize attempt")
