"""Data-parallel plumbing: rays shard across ranks (one process per GPU), the only collective is a
sum-all-reduce of the parameter gradients (reference: Lightning DDP, scripts/train_kitti.py:127-156).

On ROCm the "nccl" backend is RCCL over xGMI.  The renderer's parameters are ordinary nn.Parameters, so stock
``DistributedDataParallel`` works too; ``GradBucket`` is the lean path used by bench.py: both ResnetFC gradient
sets (43.3 MB fp32) travel as ONE flat bucket -- a single large all-reduce suits point-to-point xGMI links
better than many small per-layer messages.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist


# True: the gradient hooks below issue their collectives also in a process group of ONE rank (where they are arithmetically no-ops).
# For tests and ``bench.py --force-dist``: the RCCL code path -- communicator bound to the device, stream / event handling of
# asynchronous work, graph capture of a collective -- then runs on a single leased GPU.
FORCE_COLLECTIVES = False


def _active() -> bool:
    return dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)


def numa_pin(local_rank: int) -> Optional[str]:
    """Restrict this process to the CPUs of its GPU's NUMA node (one process per GPU: its launch calls, pinned buffers and the RCCL
    proxy thread then stay on the socket the GPU hangs off).  Best effort: returns a description of what was done, or None when the
    topology is not readable (no sysfs entry, one node only, affinity already narrower)."""
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        bus = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        with open("/sys/bus/pci/devices/%s/numa_node" % bus) as fh:
            node = int(fh.read().strip())
        if node < 0:
            return None
        with open("/sys/devices/system/node/node%d/cpulist" % node) as fh:
            cpus = set()
            for part in fh.read().strip().split(","):
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
        have = os.sched_getaffinity(0)
        want = cpus & have
        if not want or want == have:
            return None
        os.sched_setaffinity(0, want)
        return "NUMA node %d (%d CPUs) for GPU %s" % (node, len(want), bus)
    except Exception:
        return None


def more_hw_queues(n: int = 8) -> bool:
    """The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and a hardware queue runs
    its kernels in order.  The renderer overlaps three streams (the caller's, its side stream, the library's weight-gradient stream); with
    RCCL's stream in the process two of them end up on ONE queue and the step serialises -- measured with a one-rank RCCL group, eagerly
    issued: 3.016 ms per step with 4 queues, 2.646 with 8 (16: the same).  The variable is read when the runtime initialises, so this must
    run before the first GPU call of the process -- best of all before ``import torch`` (bench.py's first lines do that for its ranks):
    ``torch.cuda.is_initialized()`` is only a lower bound, ``torch.cuda.device_count()`` can already have brought the HIP runtime up while
    it still says False.  Returns False (and changes nothing) if torch's CUDA state is initialised; True means "the variable is set", which
    is a promise about the runtime only if no GPU call came first."""
    if "GPU_MAX_HW_QUEUES" in os.environ:
        return True
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        return False
    os.environ["GPU_MAX_HW_QUEUES"] = str(n)
    return True


def init_from_env(backend: Optional[str] = None, force: bool = False, graph_capture: bool = False) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from torchrun's environment; initialises the process group if world > 1 (``force``: also for
    a single rank).  ``graph_capture``: the collectives of this group will be captured into a hipGraph (GraphedStep with the gradient
    hooks installed).  ProcessGroupNCCL's watchdog thread polls the completion events of outstanding work; an event recorded inside a
    capture must not be queried (hipErrorCapturedEvent: the watchdog then terminates the process -- observed on RCCL, r04), so its
    asynchronous error handling is switched off for this process, as PyTorch's whole-network-capture recipe for DDP prescribes
    (TORCH_NCCL_ASYNC_ERROR_HANDLING=0; must be set before the group exists)."""
    if graph_capture:
        os.environ["TORCH_NCCL_ASYNC_ERROR_HANDLING"] = "0"
        os.environ["NCCL_ASYNC_ERROR_HANDLING"] = "0"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or force:
        more_hw_queues()       # (ranks of a process group only: a plain one-process run is left as the runtime comes)
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("SRF_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":   # bind the communicator to this rank's GPU up front (no device guessing in barrier())
            torch.cuda.set_device(local % torch.cuda.device_count())
            kw["device_id"] = torch.device("cuda", local % torch.cuda.device_count())
        try:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
        except TypeError:       # older torch without device_id
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


_SIDE_GROUP = None
_SIDE_GROUP_OWNER = [None]     # the default group the side group was made under (re-initialisation invalidates it)


def all_ranks_agree(ok: bool) -> bool:
    """AND of ``ok`` over the ranks, over a gloo SIDE group (CPU tensors, TCP): for decisions that must not touch the gradient
    communicator -- e.g. "did every rank's hipGraph capture succeed?" (GraphedStep.build_on_all_ranks), where a failed capture may have
    left the RCCL stream of that rank in an undefined state and a rank that failed before reaching a collective would leave the others
    waiting inside one.  Every rank must call it the same number of times.  One process: returns ``ok``."""
    global _SIDE_GROUP
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return bool(ok)
    if _SIDE_GROUP is not None and _SIDE_GROUP_OWNER[0] is not dist.group.WORLD:
        _SIDE_GROUP = None           # the process group was destroyed and re-initialised: the cached side group belongs to the old one
    if _SIDE_GROUP is None:
        if dist.get_backend() == "gloo":
            _SIDE_GROUP = dist.group.WORLD
            _SIDE_GROUP_OWNER[0] = dist.group.WORLD
        else:
            # one node (the rendezvous address is loopback): gloo on the loopback interface -- a container's hostname need not resolve
            if os.environ.get("MASTER_ADDR", "") in ("127.0.0.1", "localhost", "::1"):
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            try:
                grp = dist.new_group(backend="gloo")
            except Exception:        # noqa: BLE001 -- no gloo transport on this rank
                grp = None
            # the ranks must all take the SAME path: a rank whose gloo group failed would otherwise reduce on the default group while
            # the others wait in the gloo one.  The outcome of new_group is itself agreed on over the default group -- the
            # one exchange here that does use the gradient communicator, at set-up time, before any capture has been attempted.
            flag = torch.tensor([1 if grp is not None else 0], dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            _SIDE_GROUP = grp if int(flag[0]) == 1 else False
            _SIDE_GROUP_OWNER[0] = dist.group.WORLD
    if _SIDE_GROUP is False:
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t[0]))
    t = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=_SIDE_GROUP)
    return bool(int(t[0]))


def shard_rays(n_rays_total: int, rank: int, world: int) -> Tuple[int, int]:
    """[begin, end) of this rank's contiguous ray shard (rays are independent: no exchange on the render path)."""
    base, rem = divmod(n_rays_total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


# Optional timing of the collectives (bench.py --gpus N): when a list is installed here, every reduction appends
# (kind, start_event, end_event); kind "wait" brackets the point where the consuming stream blocks on an asynchronous reduction (the
# EXPOSED part of that all-reduce), kind "sync" brackets a reduction issued in stream order (its whole duration, on whatever stream
# it was issued on -- the gaussian head's runs on the backward's side stream, under the radiance MLP's kernels).
TIMING = None


# Number of gradient collectives this process has issued through the two hooks below since the last verify_step_collectives().
_ISSUED = 0


def verify_step_collectives() -> int:
    """The per-session hooks (``model.grad_sync`` / ``model.grad_sync_async``) issue ONE pair of all-reduces per render_rays_batch
    session whose gradients reach autograd, so every rank must open the same number of sessions per step, in the same order, and
    use each session's output in its loss -- a rank with one source frame fewer deadlocks the others.  Call this once per step
    (after backward) to turn that deadlock into an error: it compares the number of hook collectives issued since the last call
    across ranks (one 2-element all-reduce) and raises on every rank when they differ.  Returns the count.  The hooks must not be
    combined with DistributedDataParallel / Lightning gradient hooks on the same parameters (the gradients would be averaged
    twice); for a step whose session count varies between ranks leave both hooks None and reduce once per step with
    ``GradBucket(model.parameters()).allreduce_mean()`` after backward."""
    global _ISSUED
    n, _ISSUED = _ISSUED, 0
    if _active():
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor([n, -n], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        hi, lo = int(t[0]), -int(t[1])
        if hi != lo:
            raise RuntimeError("scenerf_amd.dist: ranks issued different numbers of gradient collectives this step (min %d, max %d, "
                               "this rank %d): every rank must open the same render_rays_batch sessions per step" % (lo, hi, n))
    return n


def _mark():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


_AVG_OK = None      # does the backend average inside the collective (RCCL: ncclAvg)?  None = not tried yet


def _native_avg() -> bool:
    """The mean taken INSIDE the all-reduce (``ReduceOp.AVG`` = ncclAvg on RCCL): no 21.7-MB division kernel behind each of the step's two
    collectives.  gloo has no AVG: sum, then divide."""
    global _AVG_OK
    if _AVG_OK is None:
        _AVG_OK = dist.get_backend() == "nccl" and hasattr(dist.ReduceOp, "AVG")
    return _AVG_OK


def allreduce_mean_(flat: torch.Tensor) -> None:
    """In-place mean over ranks of a flat gradient buffer (no-op for a single process).  Installed as
    ``model.grad_sync``: the renderer calls it once per MLP on the packed fp32 gradient sink (21.7 MB)."""
    global _ISSUED, _AVG_OK
    if _active():
        _ISSUED += 1
        t0 = _mark() if (TIMING is not None and flat.is_cuda) else None
        if _native_avg():
            try:
                dist.all_reduce(flat, op=dist.ReduceOp.AVG)
            except RuntimeError:      # a backend build without ncclAvg: remember, and take the mean by hand (nothing was reduced yet)
                _AVG_OK = False
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
                flat.div_(dist.get_world_size())
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat.div_(dist.get_world_size())
        if t0 is not None:
            TIMING.append(("sync", t0, _mark()))


def allreduce_mean_async(flat: torch.Tensor):
    """Start the same reduction without blocking the calling stream; returns ``finish()`` -- call it (on the stream that will read
    ``flat``) before the gradients are used -- or None for a single process.  Installed as ``model.grad_sync_async``."""
    global _ISSUED
    if not _active():
        return None
    _ISSUED += 1
    avg = _native_avg()
    work = dist.all_reduce(flat, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, async_op=True)
    world = dist.get_world_size()

    def finish():
        t0 = _mark() if (TIMING is not None and flat.is_cuda) else None
        work.wait()          # the current stream waits for the collective
        if t0 is not None:
            TIMING.append(("wait", t0, _mark()))
        if not avg:
            flat.div_(world)
    return finish


class GradBucket:
    """Flat fp32 bucket over a fixed parameter list; ``allreduce_mean`` averages .grad across ranks in one call."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.views = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def pack(self) -> None:
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)

    def unpack(self) -> None:
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)

    def allreduce_mean(self, async_op: bool = False):
        """Sum-all-reduce the bucket and divide by world size (no-op for a single process)."""
        if not _active():
            return None
        self.pack()
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)
        if async_op:
            return work
        self.finish()
        return None

    def finish(self) -> None:
        self.flat.div_(dist.get_world_size())
        self.unpack()


class StepGradSync:
    """Once-per-backward gradient averaging that does not care how many ``render_rays_batch`` sessions a rank opened.

    The per-session hooks above are the fast path for a step with ONE session per rank (bench.py: the collectives overlap the
    backward).  A real KITTI batch renders ``len(T_source2infers)`` source frames per image (scenerf.py:266-272) and that count differs
    between images, hence between ranks: with per-session collectives a rank with one frame fewer would leave the others waiting.
    This object is the safe default for such trainers when DistributedDataParallel is not used: the gradients of all sessions
    accumulate locally in ``.grad`` as usual; the first parameter gradient of a backward arms ONE end-of-backward callback (the
    mechanism DDP itself uses), which averages both MLPs' gradients in a single flat all-reduce (43.3 MB over RCCL).  Every rank
    whose backward reaches at least one of the parameters issues exactly one collective, whatever its session count; a rank whose
    step produced no gradient for any of them must call ``sync()`` itself (the others are waiting in the all-reduce).

        sync = StepGradSync(list(model.mlp.parameters()) + list(model.mlp_gaussian.parameters()))
        ... loss.backward()      # reduced when backward returns
        sync.close()             # remove the hooks
    """

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.bucket = GradBucket(params)
        self._armed_task = None      # id of the backward pass (autograd graph task) whose end-of-backward callback is queued
        self.reductions = 0
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.bucket.params]

    @staticmethod
    def _task_id() -> int:
        f = getattr(torch._C, "_current_graph_task_id", None)
        return int(f()) if f is not None else -1

    def _on_grad(self, _p) -> None:
        # armed PER BACKWARD PASS: a backward that raised never ran its callback -- a flag that only the callback resets would then
        # stay set and every later backward of this rank would skip its all-reduce silently (the other ranks hang or the replicas
        # drift apart).  The graph task's id names the pass; without that API (old torch) the flag is cleared by ``begin_step``.
        tid = self._task_id()
        if self._armed_task is None or (tid >= 0 and tid != self._armed_task):
            self._armed_task = tid
            torch.autograd.Variable._execution_engine.queue_callback(self._finish)

    def begin_step(self) -> None:
        """Forget a callback that never ran (a backward that raised).  Needed only where ``torch._C._current_graph_task_id`` is
        missing; harmless otherwise."""
        self._armed_task = None

    def _finish(self) -> None:
        self._armed_task = None
        self.reductions += 1
        self.bucket.allreduce_mean()

    def sync(self) -> None:
        """The collective of a step whose backward reached NO parameter of the bucket on this rank (no hook fired, so no callback was
        queued): every rank must take part in every step's all-reduce, so such a rank calls this once instead."""
        self.reductions += 1
        self.bucket.allreduce_mean()

    def close(self) -> None:
        for h in self._handles:
            h.remove()
        self._handles = []
