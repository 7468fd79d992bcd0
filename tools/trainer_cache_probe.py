"""Losses of the trainer's per-image step (tests/test_gpu_graph.py::_trainer_setup) over W + N steps: eager / GraphedFn, with and without the
step-scoped caches (converted maps, packed operands)."""
import contextlib, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_gpu_graph as T            # noqa: E402
from scenerf_amd.graph import GraphedFn  # noqa: E402

W, N = 2, 3
SHARE = os.environ.get('PROBE_SHARE', '1') == '1'


def setup(caches):
    m, opt, maps, batch = T._trainer_setup(31)
    if not caches:
        m.cache_converted_maps = False
        m._params_fixed = contextlib.nullcontext
    m.share_image_sessions = SHARE
    return m, opt, maps, batch


def eager(caches):
    m, opt, maps, batch = setup(caches)
    out = []
    for _ in range(W + N):
        opt.zero_grad(set_to_none=True)
        for v in maps.values():
            v.grad = None
        loss = m.step(batch, "train")
        loss.backward()
        opt.step()
        out.append(round(float(loss.detach()), 5))
    return out


def graphed(caches):
    m, opt, maps, batch = setup(caches)
    gs = GraphedFn(m, opt, lambda: m.step(batch, "train"), T.DEV, grad_leaves=list(maps.values()), warmup=W)
    return [round(float(gs()), 5) for _ in range(N)]


for c in (True, False):
    print("eager   caches=%s" % c, eager(c))
    print("eager   caches=%s" % c, eager(c))
    print("graphed caches=%s" % c, graphed(c))
    print("graphed caches=%s" % c, graphed(c))
