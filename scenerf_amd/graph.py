"""One training step of the hot path as ONE hipGraph: ``render_rays_batch`` forward, the caller's loss on its outputs, the backward into
both MLPs and the feature maps, and the optimizer step are captured once and replayed -- one launch per step instead of ~60, so the
step costs what its kernels cost whatever the host is doing (the eager step is issued in ~2.0 ms of host time against ~2.6 ms of GPU
time on an idle host; a busy host makes it host-bound).

What a captured step fixes, as with any CUDA / HIP graph: tensor ADDRESSES and shapes.  Camera, pose and pixels are read from the
static tensors ``cam_K`` / ``T_source2infer`` / ``pixels`` of the object (``copy_`` new values into them between replays); the feature
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
constructor returns (``steps_warmup``); pass ``restore=True`` to have parameters, optimizer state and RNG state put back afterwards,
so that the first replay is step 1 on the data the caller copies into the static tensors.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch


class GraphedStep:
    def __init__(self, model, optimizer: Optional[torch.optim.Optimizer], loss_fn: Callable[[Dict[str, torch.Tensor]], torch.Tensor],
                 cam_K: torch.Tensor, T_source2infer: torch.Tensor, x_rgb: Dict[str, torch.Tensor], pixels: torch.Tensor,
                 ray_batch_size: Optional[int] = None, warmup: int = 3, noise=None, restore: bool = False):
        """``noise``: optional static ``(noise_u, noise_g)`` tensors handed to ``render_rays_batch`` (the caller refills them between
        replays); without it the sampler draws on the device inside the graph.  ``restore``: undo the warm-up steps' effect on
        parameters, optimizer state and the device RNG (module docstring)."""
        if noise is None and not getattr(model.render_cfg, "device_rng", False):
            raise RuntimeError("GraphedStep: the sampler noise must be drawn on the device (render_cfg.device_rng = True); the "
                               "reference's host-side draw cannot be captured")
        for g in (optimizer.param_groups if optimizer is not None else ()):
            if not g.get("capturable", False):
                raise RuntimeError("GraphedStep: the optimizer must be capturable (FusedAdamW(capturable=True))")
        dev = pixels.device
        self.model, self.optimizer, self.loss_fn = model, optimizer, loss_fn
        self.cam_K, self.T_source2infer, self.pixels, self.x_rgb = cam_K, T_source2infer, pixels, x_rgb
        self.ray_batch_size = int(ray_batch_size or pixels.shape[0])
        self.noise = noise
        self._one = None
        self._params = [p for g in optimizer.param_groups for p in g["params"]] if optimizer is not None else \
            [p for p in model.parameters() if p.requires_grad]
        self._map_leaves = [v for v in x_rgb.values() if v.requires_grad]
        snap = None
        if restore:
            snap = ([p.detach().clone() for p in self._params], torch.cuda.get_rng_state(dev))
        # warm-up on a side stream (allocator pools, one-time setup, optimizer state), as torch's capture recipe asks
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._eager()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl":
            # the warm-up steps' collectives are still on ProcessGroupNCCL's watchdog list until its next sweep; a completion query that
            # thread makes while this one is capturing is an error inside the watchdog (observed on RCCL, ~1 run in 10: the process
            # aborts).  All of that work has finished (synchronize above): give the sweep time to retire it before the capture starts
            import time
            time.sleep(0.6)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._eager()
        self.map_grads = {k: v.grad for k, v in x_rgb.items() if v.requires_grad}
        self.steps_warmup = max(1, warmup)
        if snap is not None:
            # parameters and moments back IN PLACE (the graph holds their addresses), step counts to zero, the RNG where it was
            with torch.no_grad():
                for p, v in zip(self._params, snap[0]):
                    p.copy_(v)
                if optimizer is not None:
                    for st in optimizer.state.values():
                        for k, v in st.items():
                            if torch.is_tensor(v):
                                v.zero_()
            torch.cuda.set_rng_state(snap[1], dev)
            self.steps_warmup = 0

    def _eager(self) -> torch.Tensor:
        for p in self._params:
            p.grad = None
        for v in self._map_leaves:
            v.grad = None
        out = self.model.render_rays_batch(self.cam_K, self.T_source2infer, self.x_rgb, T_cam2velo=None, sampled_pixels=self.pixels,
                                           ray_batch_size=self.ray_batch_size, **({"noise": self.noise} if self.noise is not None else {}))
        loss = self.loss_fn(out)
        if self._one is None or self._one.shape != loss.shape or self._one.dtype != loss.dtype:
            self._one = torch.ones_like(loss)
        loss.backward(self._one)        # (a cached root gradient: autograd's own ones_like is one more fill launch per step)
        if self.optimizer is not None:
            self.optimizer.step()
        return loss.detach()

    def __call__(self) -> torch.Tensor:
        """Replay: one more training step.  Returns the (static) loss tensor of the captured step."""
        if self.optimizer is not None and hasattr(self.optimizer, "sync_hyper"):
            self.optimizer.sync_hyper()     # a scheduler may have moved the learning rate
        self.graph.replay()
        return self.loss
