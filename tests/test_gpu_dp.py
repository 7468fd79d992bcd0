"""Data-parallel gradient path on a GPU: two ranks over gloo on the one GPU of the test box.  Covers what the CPU gloo
test cannot: the renderer's packed gradient sinks, the head's early all-reduce on the side stream, and the `synced` hand-off."""
import os
import re
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_two_ranks_average_their_gradients():
    env = dict(os.environ, SRF_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(HERE, "dp_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    err = re.search(r"DP_ERROR.*?(?=DP_ERROR|\Z)", r.stdout, re.S)
    assert r.returncode == 0, (err.group(0)[-3000:] if err else r.stdout[-1500:] + r.stderr[-1500:])
    m = re.search(r"DP_RESULT same=(\w+) rel=([\d.e+-]+) local_vs_mean=([\d.e+-]+) ddp=([\d.e+-]+)", r.stdout)
    assert m, r.stdout[-2000:] + r.stderr[-2000:]
    assert m.group(1) == "True"                 # every rank ends up with the same gradient
    assert float(m.group(2)) < 1e-3             # ... which is the mean of the per-rank gradients
    assert float(m.group(3)) > 1e-2             # (and the shards' own gradients do differ)
    assert float(m.group(4)) < 1e-3             # stock DistributedDataParallel around the same step gives the same mean


@pytest.mark.gpu
def test_two_ranks_at_the_benched_shape_fall_back_from_the_capture_together_and_still_average():
    """tests/dp_worker.py::main_benched_shape_and_capture: R = 1,200 x N = 128 per rank on the full KITTI sphere, `build_on_all_ranks(
    GraphedStep)` over gloo (not capturable: both ranks must drop to the eager step together), then synced == mean of local on every rank."""
    env = dict(os.environ, SRF_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29537", os.path.join(HERE, "dp_worker.py"), "benched"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    err = re.search(r"DP_ERROR.*?(?=DP_ERROR|\Z)", r.stdout, re.S)
    assert r.returncode == 0, (err.group(0)[-3000:] if err else r.stdout[-1500:] + r.stderr[-1500:])
    m = re.search(r"DP_BENCHED fallback_on_all=(\w+) same=(\w+) rel=([\d.e+-]+) local_vs_mean=([\d.e+-]+)", r.stdout)
    assert m, r.stdout[-2000:] + r.stderr[-2000:]
    assert m.group(1) == "True"                 # no rank kept a graph: the rank-consistent fallback fired everywhere
    assert m.group(2) == "True" and float(m.group(3)) < 1e-3 and float(m.group(4)) > 1e-2
