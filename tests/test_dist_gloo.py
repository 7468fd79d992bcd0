"""N>1 path on CPU: two gloo processes shard the rays and average a gradient bucket exactly like bench.py does
over RCCL (rays are independent; the only collective is the parameter-gradient all-reduce)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from scenerf_amd import dist as sdist
    r, w, _ = sdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    lin = torch.nn.Linear(7, 5)
    with torch.no_grad():
        for p in lin.parameters():
            p.fill_(1.0)
    # each rank "renders" its shard of 10 rays and produces shard-dependent grads
    b, e = sdist.shard_rays(10, rank, world)
    x = torch.arange(b, e, dtype=torch.float32).reshape(-1, 1).repeat(1, 7)
    lin(x).sum().backward()
    local = [p.grad.clone() for p in lin.parameters()]
    bucket = sdist.GradBucket(lin.parameters())
    bucket.allreduce_mean()
    flat = torch.full((5,), float(rank + 1))
    sdist.allreduce_mean_(flat)                       # the hook bench.py installs as model.grad_sync
    assert torch.allclose(flat, torch.full((5,), (1 + world) / 2.0))
    flat2 = torch.full((5,), float(10 * (rank + 1)))
    finish = sdist.allreduce_mean_async(flat2)        # ... and its early form (model.grad_sync_async): start now, finish later
    assert callable(finish)
    finish()
    assert torch.allclose(flat2, torch.full((5,), 10 * (1 + world) / 2.0))
    assert sdist.verify_step_collectives() == 2       # both ranks issued the same two hook collectives this "step"
    if rank == 0:                                     # a rank that opens one session more: an error on every rank, not a hang
        sdist._ISSUED += 1
    with pytest.raises(RuntimeError, match="different numbers of gradient collectives"):
        sdist.verify_step_collectives()
    torch.save(dict(local=local, reduced=[p.grad.clone() for p in lin.parameters()], shard=(b, e)), out % rank)
    dist.destroy_process_group()


def test_two_process_gloo_allreduce(tmp_path):
    world = 2
    out = str(tmp_path / "r%d.pt")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    res = [torch.load(out % r) for r in range(world)]
    assert res[0]["shard"] == (0, 5) and res[1]["shard"] == (5, 10)
    for i in range(2):
        mean = (res[0]["local"][i] + res[1]["local"][i]) / 2
        for r in range(world):
            torch.testing.assert_close(res[r]["reduced"][i], mean)
    assert not torch.equal(res[0]["local"][0], res[1]["local"][0])


def _worker_step_sync(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from scenerf_amd import dist as sdist
    sdist.init_from_env(backend="gloo")
    lin = torch.nn.Linear(3, 2)
    with torch.no_grad():
        for p in lin.parameters():
            p.fill_(0.5)
    sync = sdist.StepGradSync(lin.parameters())
    # rank 0 renders two "source frames" this step, rank 1 one (KITTI batches differ in len(T_source2infers)): the per-session hooks
    # would hang here; StepGradSync reduces once per backward on every rank
    n_sessions = 2 if rank == 0 else 1
    loss = sum(lin(torch.full((4, 3), float(rank + 1 + s))).sum() for s in range(n_sessions))
    local = torch.autograd.grad(loss, list(lin.parameters()), retain_graph=True)
    loss.backward()
    assert sync.reductions == 1
    sync.close()
    torch.save(dict(local=[g.clone() for g in local], reduced=[p.grad.clone() for p in lin.parameters()]), out % rank)
    dist.destroy_process_group()


def test_step_sync_with_unequal_session_counts(tmp_path):
    world = 2
    out = str(tmp_path / "s%d.pt")
    mp.spawn(_worker_step_sync, args=(world, _free_port(), out), nprocs=world, join=True)
    res = [torch.load(out % r) for r in range(world)]
    for i in range(2):
        mean = (res[0]["local"][i] + res[1]["local"][i]) / 2
        for r in range(world):
            torch.testing.assert_close(res[r]["reduced"][i], mean)


def test_shard_rays_covers_everything():
    from scenerf_amd.dist import shard_rays
    for n in (1, 7, 1200, 1201):
        for w in (1, 2, 3, 8):
            spans = [shard_rays(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1


def test_bench_control_flow_over_gloo():
    """bench.py --dry-run under torch.distributed.run with two gloo ranks: the bench's own control flow (which rank issues which
    collective when, the hooks removed before the rank-0-only legs, the per-rank all-reduce report, one JSON line, clean exit) around
    a stub step.  A rank-0-only leg that issued a collective -- the bug fixed in round 1 -- hangs here and fails by timeout."""
    import json
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["value"] > 0
    assert len(d["allreduce"]["per_rank_ms_per_step"]) == 2
    alone = d["allreduce"]["standalone"]          # the same collective outside the step, per rank (what a first N-GPU run reads first)
    assert len(alone["us_per_collective_per_rank"]) == 2 and all(u > 0 for u in alone["us_per_collective_per_rank"]) and alone["bytes"] == 4096


def test_bench_falls_back_on_every_rank_when_one_capture_fails():
    """The rank-consistent fallback of the captured step (scenerf_amd.graph.build_on_all_ranks) in bench.py's own control flow: rank 1's
    (stub) capture fails, rank 0's succeeds -- both must step eagerly (the line says so), the collectives still pair up (no hang), one
    JSON line, clean exit.  And without the failure the line reports the replayed step."""
    import json
    import subprocess
    for fail, want in ((1, "eager (capture failed on another rank"), (0, "eager (capture failed on this rank"), (-1, "one hipGraph replay per step")):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1",
               "--graph", "on", "--dry-fail-capture", str(fail)]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["value"] > 0
        assert d["config"]["step_issue"].startswith(want), d["config"]["step_issue"]
        if fail == -1:     # the replayed collectives were checked before the timed region (bench.replay_agrees_across_ranks)
            assert "bit-equal across ranks" in d["config"]["step_issue"], d["config"]["step_issue"]


def test_bench_steps_eagerly_on_more_than_one_rank_unless_asked():
    """`--graph auto` (the default) with two ranks: the eager step, and the line says why (a replayed all-reduce between GPUs has never run:
    opt-in with --graph on until it has -- ADVICE r05)."""
    import json
    r = _bench("--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["config"]["step_issue"].startswith("eager (more than one rank"), d["config"]["step_issue"]


def test_replay_check_sees_ranks_that_drifted_apart(tmp_path):
    """bench.replay_agrees_across_ranks: equal parameters pass, one rank one ulp off fails -- on EVERY rank."""
    out = str(tmp_path / "r%d.pt")
    mp.spawn(_worker_replay_check, args=(2, _free_port(), out), nprocs=2, join=True)
    for r in range(2):
        ok_same, ok_diff = torch.load(out % r)
        assert ok_same[0] is True and ok_same[1] == 0.0
        assert ok_diff[0] is False and ok_diff[1] > 0.0


def _worker_replay_check(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from scenerf_amd import dist as sdist
    import bench
    sdist.init_from_env(backend="gloo")
    g = torch.Generator().manual_seed(5)
    ps = [torch.randn(512, 512, generator=g), torch.randn(37, generator=g)]
    same = bench.replay_agrees_across_ranks(ps, world)
    if rank == 1:
        ps[1][3] = torch.nextafter(ps[1][3], torch.tensor(10.0))
    diff = bench.replay_agrees_across_ranks(ps, world)
    torch.save((same, diff), out % rank)
    dist.destroy_process_group()


def _worker_agree(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from scenerf_amd import dist as sdist
    from scenerf_amd.graph import build_on_all_ranks
    sdist.init_from_env(backend="gloo")
    res = []
    assert sdist.all_ranks_agree(True) is True
    assert sdist.all_ranks_agree(rank == 0) is False          # one dissenting rank: False on EVERY rank

    def factory_ok():
        return "graph of rank %d" % rank

    def factory_fail_on_1():
        if rank == 1:
            raise RuntimeError("hipErrorStreamCaptureInvalidated (pretend)")
        return "graph of rank %d" % rank

    g, note = build_on_all_ranks(factory_ok)
    res.append((g, note))
    g, note = build_on_all_ranks(factory_fail_on_1)
    res.append((g, note))
    t = torch.tensor([float(rank)])                             # the default group still works after the side group's traffic
    dist.all_reduce(t)
    res.append(float(t))
    torch.save(res, out % rank)
    dist.destroy_process_group()


def test_capture_outcome_is_agreed_over_a_side_group(tmp_path):
    world = 2
    out = str(tmp_path / "a%d.pt")
    mp.spawn(_worker_agree, args=(world, _free_port(), out), nprocs=world, join=True)
    res = [torch.load(out % r) for r in range(world)]
    for r in range(world):
        assert res[r][0] == ("graph of rank %d" % r, "captured on every rank")
        assert res[r][1][0] is None                              # nobody keeps a graph
        assert res[r][2] == 1.0
    assert "on another rank" in res[0][1][1] and "on this rank" in res[1][1][1] and "pretend" in res[1][1][1]


def _bench(*argv, env=None, timeout=240):
    import subprocess
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True, timeout=timeout,
                          cwd=ROOT, env=e)


def test_bench_starts_its_own_ranks():
    """Plain ``python bench.py --gpus 2`` (no torchrun, no WORLD_SIZE): bench.py must start two ranks itself, like the reference's
    trainer does (train_kitti.py:127-156), and say who took part."""
    import json
    r = _bench("--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2
    assert d["ranks"]["ranks_seen"] == 2 and d["ranks"]["backend"] == "gloo"
    assert sorted(x["rank"] for x in d["ranks"]["devices"]) == [0, 1]
    assert len({x["pid"] for x in d["ranks"]["devices"]}) == 2
    assert "self-launch" in d["ranks"]["launched_by"]


def test_bench_refuses_a_world_size_mismatch():
    """--gpus N under a launcher that started M != N ranks: no JSON line, non-zero exit (a line saying n_gpus = M would be read as an
    N-GPU measurement)."""
    r = _bench("--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "refusing to measure" in r.stderr


def test_bench_restarts_a_child_the_runtime_aborted_and_says_so(tmp_path):
    """bench.py at one GPU runs the measurement in a child process and starts it again (at most twice) when the child dies of SIGABRT --
    what the HSA runtime does to a process on a GPU fault (round 6: an intermittent aperture violation on some boxes).  The line of the
    run that finished says how often; a child that always dies ends the bench with a non-zero exit and no line."""
    import json
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry-run", "--steps", "2", "--warmup", "1"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(SRF_BENCH_SUPERVISE="1", SRF_BENCH_TEST_ABORT=str(tmp_path / "aborted_once"))
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["process_restarts"] == 1 and "SIGABRT" in d["process_restart_reasons"][0] and "starting the measurement again" in r.stderr
    env["SRF_BENCH_TEST_ABORT"] = "always"
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=ROOT, env=env)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.stderr.count("starting the measurement again") == 2
    env.pop("SRF_BENCH_TEST_ABORT")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=ROOT, env=env)      # the ordinary case: no restart, no field
    assert r.returncode == 0 and "process_restarts" not in json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
