"""One training step of the hot path as ONE hipGraph: ``render_rays_batch`` forward, the caller's loss on its outputs, the backward into
both MLPs and the feature maps, and the optimizer step are captured once and replayed -- one launch per step instead of ~60, so the
step costs what its kernels cost whatever the host is doing (the eager step is issued in ~2.0 ms of host time against ~2.6 ms of GPU
time on an idle host; a busy host makes it host-bound).

What a captured step fixes, as with any CUDA / HIP graph: tensor ADDRESSES and shapes.  Camera, pose and pixels are read from the
static tensors ``cam_K`` / ``T_source2infer`` / ``pixels`` of the object (``copy_`` new values into them between replays -- the
host-made inverse of ``cam_K`` is refreshed in place by ``__call__`` when ``cam_K``'s version counter has moved; or hand
``pixels`` over as a callable that draws them on the device inside the step, like the reference's per-step ``randperm``); the feature
maps are the caller's tensors, captured by address (write new features into the same storage); their gradients land in
``map_grads`` (the ``.grad`` of the captured leaves, zeroed by the graph itself at the top of each replay).  The gaussian sampler's
noise must be drawn on the device (``RenderConfig.device_rng``: the generator state advances inside the graph); the optimizer must be
capturable (``scenerf_amd.optim.FusedAdamW(capturable=True)`` or torch's ``capturable=True`` optimizers).

One GraphedStep per set of parameters and process at a time: autograd binds a leaf's AccumulateGrad node to the stream of its first use,
so a SECOND capture over the same parameters meets nodes that belong to the first capture's stream; the cross-stream hand-off autograd
inserts is not capturable and hipStreamEndCapture fails hard (observed: a segfault, r04).  Drop the first object (and anything that keeps
its autograd graph alive) before building another.

Construction has side effects: it runs ``warmup`` real optimizer steps on whatever the static inputs hold at that moment (allocator
pools, one-time setup and the optimizer state must exist before the capture) and then one more while capturing, whose kernels do
not execute.  Parameters, AdamW moments, the step count and the device RNG have therefore advanced by ``warmup`` steps when the
constructor returns (``steps_warmup``); pass ``restore=True`` to have all of that put back afterwards IN PLACE (the graph holds the
addresses): parameters, every tensor of ``optimizer.state`` (moments and step counts as they were -- an optimizer resumed from a
checkpoint or stepped before keeps its state; entries the warm-up created are zeroed), ``FusedAdamW``'s device-side [lr, t], the model's
in-kernel sampler state ``{seed, calls}`` (``model._device_rng_state``: created before the snapshot, so the seed is the one the first
replay uses), torch's CUDA generator, and any tensor listed in ``restore_tensors`` (e.g. the ``[seed, calls]`` tensor of a loss that calls
``source_loss(rng_state=...)``).  The first replay is then step 1 from the caller's state: ``tests/test_gpu_graph.py`` holds it equal to an
eager first step.

More than one rank: ``GraphedStep.build_on_all_ranks`` attempts the capture on every rank and lets the ranks AGREE on the outcome over a
gloo side group -- if any rank's capture failed, every rank drops its graph and steps eagerly (a step captured on some ranks only would
still match the others' collectives one for one, but a bench line or a training log must not describe two kinds of rank).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Sequence

import torch


def _wait_for_pending_collectives(timeout_s: float = 10.0) -> None:
    """Block until the process group's watchdog has retired every finished collective.  The warm-up steps' collectives stay on
    ProcessGroupNCCL's work list until its next sweep; a completion query that thread makes while this one is capturing is an error inside
    the watchdog (observed on RCCL, ~1 run in 10: the process aborts).  All of that work HAS finished (the caller synchronised the device):
    ask the backend to report an empty list (ProcessGroup._wait_for_pending_works: it returns when the watchdog's list is empty) instead of
    hoping that a fixed sleep outlasts one sweep; without that API, poll the group's sequence of sweeps with a bounded sleep loop."""
    import time
    dist = torch.distributed
    if not (dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"):
        return
    pg = dist.group.WORLD
    f = getattr(pg, "_wait_for_pending_works", None)
    if f is not None:
        try:
            f()
            return
        except Exception:      # (a backend that does not implement it raises: fall through)
            pass
    t0 = time.time()           # older torch: the watchdog sweeps every 100 ms; wait for several sweeps
    while time.time() - t0 < min(timeout_s, 1.0):
        time.sleep(0.1)


OPTIMIZER_BESIDE_MAP_GRADS = True     # (module knob for A/B runs: bench.py --set graph.OPTIMIZER_BESIDE_MAP_GRADS=False)


class GraphedStep:
    def __init__(self, model, optimizer: Optional[torch.optim.Optimizer], loss_fn: Callable[[Dict[str, torch.Tensor]], torch.Tensor],
                 cam_K: torch.Tensor, T_source2infer: torch.Tensor, x_rgb: Dict[str, torch.Tensor], pixels: torch.Tensor,
                 ray_batch_size: Optional[int] = None, warmup: int = 3, noise=None, restore: bool = False,
                 restore_tensors: Sequence[torch.Tensor] = ()):
        """``noise``: optional static ``(noise_u, noise_g)`` tensors handed to ``render_rays_batch`` (the caller refills them between
        replays); without it the sampler draws on the device inside the graph.  ``restore``: undo the warm-up steps' effect on
        parameters, optimizer state and every RNG state (module docstring); ``restore_tensors``: further device tensors the step
        advances, put back with them."""
        if noise is None and not getattr(model.render_cfg, "device_rng", False):
            raise RuntimeError("GraphedStep: the sampler noise must be drawn on the device (render_cfg.device_rng = True); the "
                               "reference's host-side draw cannot be captured")
        for g in (optimizer.param_groups if optimizer is not None else ()):
            if not g.get("capturable", False):
                raise RuntimeError("GraphedStep: the optimizer must be capturable (FusedAdamW(capturable=True))")
        if (getattr(model, "grad_sync", None) is not None or getattr(model, "grad_sync_async", None) is not None) \
                and torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_backend() != "nccl":
            # refused BEFORE anything is captured (a capture that fails half-way is expensive to recover from): only RCCL's collectives
            # are stream operations; gloo's stage through the host
            raise RuntimeError("GraphedStep: gradient collectives over the %r backend cannot be captured (they stage through the host); "
                               "only nccl (= RCCL) collectives are stream operations" % torch.distributed.get_backend())
        # ``pixels``: a static (R, 2) tensor, or a zero-argument callable evaluated INSIDE the captured step (e.g. the reference's per-step draw
        # ``grid[torch.randperm(len(grid), device=dev)[:R]]``, scenerf.py:253-264: torch's CUDA generator is graph-safe, every replay
        # draws fresh pixels); the callable must return the same shape on the same device every time and launch only capturable work
        self._pixels_fn = pixels if callable(pixels) else None
        self.model, self.optimizer = model, optimizer
        self.noise = noise
        self._params = [p for g in optimizer.param_groups for p in g["params"]] if optimizer is not None else \
            [p for p in model.parameters() if p.requires_grad]
        # (restore=True: the snapshot -- torch's CUDA generator included -- is taken BEFORE the probe draw below, so the first replay's
        # pixels are the caller's generator state's, as an eager first step's would be)
        snap = self._snapshot(cam_K.device, restore_tensors) if restore else None
        if self._pixels_fn is not None:
            pixels = self._pixels_fn()
        dev = pixels.device
        self.loss_fn = loss_fn
        self.cam_K, self.T_source2infer, self.pixels, self.x_rgb = cam_K, T_source2infer, pixels, x_rgb
        self.ray_batch_size = int(ray_batch_size or pixels.shape[0])
        self._one = None
        self._opt_stream = None
        self._map_leaves = [v for v in x_rgb.values() if v.requires_grad]
        self._capture(dev, warmup, snap)
        self.map_grads = {k: v.grad for k, v in x_rgb.items() if v.requires_grad}
        # the captured step reads the inverse intrinsics from the model's cached tensor (made on the host, SceneRF._inv_K): remember the
        # version of cam_K it belongs to -- __call__ refreshes that tensor in place when the caller has copied new intrinsics in
        self._cam_K_version = cam_K._version

    def _capture(self, dev, warmup, snap):
        """Warm-up steps, then the capture of one more (``self._eager``); ``snap``: a ``_snapshot`` to put back afterwards (restore=True)."""
        # warm-up on a side stream (allocator pools, one-time setup, optimizer state), as torch's capture recipe asks
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._eager()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        _wait_for_pending_collectives()
        self.graph = torch.cuda.CUDAGraph()
        cur = torch.cuda.current_stream(dev)
        try:
            with torch.cuda.graph(self.graph):
                self.loss = self._eager()
        except BaseException:
            # a step that cannot be captured (a host sync inside it, a collective that stages through the host ...): leave no capture
            # behind.  torch.cuda.graph's exit ends the capture, but when THAT raises too (it does for an invalidated capture) the
            # capture stream stays current -- and on ROCm every later launch of the process then fails with "operation failed due to a
            # previous error during capture" (tests/test_gpu_dp.py's benched-shape worker, round 6).
            try:
                self.graph.capture_end()
            except Exception:            # noqa: BLE001 -- already ended
                pass
            torch.cuda.set_stream(cur)
            # ... and the streams that were forked into the aborted capture are abandoned: the renderer's side stream fails its next launch
            # with "invalid argument" (tools/capture_abort_probe.py: with fresh ones the eager step runs again)
            from . import renderer
            renderer.reset_side_streams()
            getattr(self.model, "__dict__", {}).pop("_metric_streams", None)      # (training.TrainingMixin's second stream, forked into it too)
            self._opt_stream = None
            self.graph = None
            raise
        self.steps_warmup = max(1, warmup)
        if snap is not None:
            self._restore(snap, dev)
            self.steps_warmup = 0

    # ---- restore=True: everything the warm-up steps advanced, put back in place -----------------------------------------------
    def _snapshot(self, dev, extra):
        opt = self.optimizer
        rng_model = None
        if self.noise is None and hasattr(self.model, "_device_rng_state"):
            rng_model = self.model._device_rng_state(dev)          # (created now if need be: the seed the replays will use)
        st = {}
        if opt is not None:
            for p, d in opt.state.items():
                st[p] = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in d.items()}
        hyper = {gi: h[0].detach().clone() for gi, h in getattr(opt, "_hyper", {}).items()} if opt is not None else {}
        more = [t for t in getattr(self.model, "_rng_states", {}).values() if t is not rng_model and t.device == torch.device(dev)]
        return dict(params=[p.detach().clone() for p in self._params], opt_state=st, hyper=hyper,
                    cuda_rng=torch.cuda.get_rng_state(dev), rng_model=(rng_model, rng_model.clone()) if rng_model is not None else None,
                    extra=[(t, t.detach().clone()) for t in list(extra) + more])

    def _restore(self, snap, dev):
        opt = self.optimizer
        with torch.no_grad():
            for p, v in zip(self._params, snap["params"]):
                p.copy_(v)
            if opt is not None:
                for p, d in opt.state.items():
                    was = snap["opt_state"].get(p, {})
                    for k, v in list(d.items()):
                        if torch.is_tensor(v):
                            if k in was and torch.is_tensor(was[k]):
                                v.copy_(was[k])
                            elif k in was:
                                v.fill_(float(was[k]))
                            else:
                                v.zero_()               # state the warm-up created: as if it had never stepped
                        elif k in was:
                            d[k] = was[k]
                        elif isinstance(v, (int, float)):
                            d[k] = type(v)(0)
                for gi, h in getattr(opt, "_hyper", {}).items():        # FusedAdamW's device-side [lr, t, scratch]
                    if gi in snap["hyper"]:
                        h[0].copy_(snap["hyper"][gi])
                    else:
                        steps = [float(s["step"]) for s in (snap["opt_state"].get(p) for p in opt.param_groups[gi]["params"]) if s and "step" in s]
                        h[0][1:].zero_()
                        h[0][1:2].fill_(max(steps) if steps else 0.0)
            if snap["rng_model"] is not None:
                snap["rng_model"][0].copy_(snap["rng_model"][1])
            for t, v in snap["extra"]:
                t.copy_(v)
        torch.cuda.set_rng_state(snap["cuda_rng"], dev)

    # ---- more than one rank: every rank captures, then all agree ----------------------------------------------------------------
    @classmethod
    def build_on_all_ranks(cls, *args, agree: Optional[Callable[[bool], bool]] = None, **kw):
        """(GraphedStep or None, note): ``build_on_all_ranks(lambda: cls(*args, **kw), agree)`` (module function below)."""
        return build_on_all_ranks(lambda: cls(*args, **kw), agree)

    def _eager(self) -> torch.Tensor:
        for p in self._params:
            p.grad = None
        for v in self._map_leaves:
            v.grad = None
        if self._pixels_fn is not None:
            self.pixels = self._pixels_fn()      # (captured: the draw is part of the replayed step; ``self.pixels`` = the last replay's)
        out = self.model.render_rays_batch(self.cam_K, self.T_source2infer, self.x_rgb, T_cam2velo=None, sampled_pixels=self.pixels,
                                           ray_batch_size=self.ray_batch_size, **({"noise": self.noise} if self.noise is not None else {}))
        loss = self.loss_fn(out)
        if self._one is None or self._one.shape != loss.shape or self._one.dtype != loss.dtype:
            self._one = torch.ones_like(loss)
        ev_box = getattr(self.model, "_step_events", None)
        if ev_box is not None:
            ev_box.pop("param_grads_ready", None)
        loss.backward(self._one)        # (a cached root gradient: autograd's own ones_like is one more fill launch per step)
        if self.optimizer is not None:
            ev = ev_box.get("param_grads_ready") if (ev_box is not None and OPTIMIZER_BESIDE_MAP_GRADS) else None
            if ev is not None and getattr(self.model, "grad_sync", None) is None and getattr(self.model, "grad_sync_async", None) is None:
                # the renderer left an event behind which every PARAMETER gradient is complete, while the feature-map gradients' tail
                # (the coarse levels' scatter: ~75 us at KITTI) still runs on its side stream: the optimizer step goes to a stream of its
                # own behind that event alone and runs beside that tail; this stream joins it before the step ends.  (Data parallel: the
                # gradient all-reduces end in PackMLP.backward on this stream -- the step stays in stream order.)
                main = torch.cuda.current_stream(self.pixels.device)
                if self._opt_stream is None:
                    self._opt_stream = torch.cuda.Stream(device=self.pixels.device)
                self._opt_stream.wait_event(ev)
                with torch.cuda.stream(self._opt_stream):
                    self.optimizer.step()
                main.wait_stream(self._opt_stream)
            else:
                self.optimizer.step()
        return loss.detach()

    def __call__(self) -> torch.Tensor:
        """Replay: one more training step.  Returns the (static) loss tensor of the captured step."""
        if self.optimizer is not None and hasattr(self.optimizer, "sync_hyper"):
            self.optimizer.sync_hyper()     # a scheduler may have moved the learning rate
        if self.cam_K is not None and self.cam_K._version != self._cam_K_version:
            # new intrinsics were copied into the static cam_K: their inverse is host-made and lives in a tensor the graph holds by address
            # (a replay runs no Python, so without this the rays would keep the OLD inverse) -- refreshed in place, outside the graph
            if hasattr(self.model, "_inv_K"):
                self.model._inv_K(self.cam_K)
            self._cam_K_version = self.cam_K._version
        self.graph.replay()
        return self.loss


class GraphedFn(GraphedStep):
    """ONE captured optimizer step of an arbitrary step function ``fn() -> loss`` -- e.g. the trainer's own per-image step
    ``lambda: model.step(batch, "train")`` (scenerf.py:119-241: per source frame a trained render, a metric-only render under no_grad and
    the source's loss, then ONE optimizer step), with several ``render_rays_batch`` sessions inside.  Everything ``fn`` reads is captured by
    ADDRESS (write new images / poses / features into the same tensors between replays); what it draws must be drawn on the device
    (``model.device_pixel_draw = True`` for the per-source pixel subset, ``render_cfg.device_rng`` for the sampler noise, the fused source
    loss's in-kernel tie-breaking noise); the intrinsics tensors must be persistent objects (a list of tensors, not slices made per call:
    their host-made inverses are cached per tensor object).  ``grad_leaves``: further leaves whose ``.grad`` the step produces (the
    feature maps of a stub encoder); they are reset to None at the top of every step like the parameters'.  The optimizer step stays in
    stream order here: with more than one session per step the parameter gradients are accumulated by autograd after each session's own."""

    def __init__(self, model, optimizer: torch.optim.Optimizer, fn: Callable[[], torch.Tensor], device, grad_leaves: Sequence[torch.Tensor] = (),
                 warmup: int = 3, restore: bool = False, restore_tensors: Sequence[torch.Tensor] = ()):
        if not getattr(model.render_cfg, "device_rng", False):
            raise RuntimeError("GraphedFn: the sampler noise must be drawn on the device (render_cfg.device_rng = True)")
        for g in optimizer.param_groups:
            if not g.get("capturable", False):
                raise RuntimeError("GraphedFn: the optimizer must be capturable (FusedAdamW(capturable=True))")
        if (getattr(model, "grad_sync", None) is not None or getattr(model, "grad_sync_async", None) is not None) \
                and torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_backend() != "nccl":
            raise RuntimeError("GraphedFn: gradient collectives over the %r backend cannot be captured" % torch.distributed.get_backend())
        dev = torch.device(device)
        self.model, self.optimizer, self._fn = model, optimizer, fn
        self.noise, self._pixels_fn, self.cam_K, self.pixels = None, None, None, None
        self._params = [p for g in optimizer.param_groups for p in g["params"]]
        self._map_leaves = list(grad_leaves)
        self._one, self._opt_stream = None, None
        snap = self._snapshot(dev, restore_tensors) if restore else None
        self._capture(dev, warmup, snap)

    def _eager(self) -> torch.Tensor:
        for p in self._params:
            p.grad = None
        for v in self._map_leaves:
            v.grad = None
        loss = self._fn()
        if self._one is None or self._one.shape != loss.shape or self._one.dtype != loss.dtype:
            self._one = torch.ones_like(loss)
        loss.backward(self._one)
        self.optimizer.step()
        return loss.detach()


def build_on_all_ranks(factory: Callable[[], object], agree: Optional[Callable[[bool], bool]] = None):
    """(graphed step or None, note).  Every rank calls this; each attempts its capture (``factory()``: normally ``lambda:
    GraphedStep(...)``); ``agree(ok)`` returns the AND over ranks (default: ``scenerf_amd.dist.all_ranks_agree``, a gloo side group -- never
    the gradient communicator, whose stream a failed capture may have left in an undefined state).  If any rank failed, EVERY rank drops
    its graph and the caller steps eagerly everywhere: no rank is left replaying a graph whose collectives the others issue by hand, and a
    log line describes one kind of rank.  (What this cannot catch: a rank that dies or hangs inside its warm-up steps -- those are real
    collectives, and the others wait in them until the process group's timeout.)"""
    from . import dist as sdist
    err = None
    try:
        g = factory()
    except Exception as e:           # noqa: BLE001 -- whatever went wrong, the other ranks must hear about it
        g, err = None, repr(e)[:200]
        try:
            if torch.cuda.is_available() and torch.cuda.is_initialized():
                torch.cuda.synchronize()
                # the failed capture's error stays behind as the HIP runtime's "last error": the library's next entry point would report
                # it for a launch that succeeded (found by tests/test_gpu_dp.py's benched-shape worker, round 6) -- cleared here
                from . import _capi
                _capi.load().scenerf_hip_clear_last_error()
        except Exception:            # noqa: BLE001
            pass
    ok = (agree or sdist.all_ranks_agree)(g is not None)
    if ok:
        return g, "captured on every rank"
    if g is not None:
        del g
        return None, "capture failed on another rank: all ranks step eagerly"
    return None, "capture failed on this rank (%s): all ranks step eagerly" % err
