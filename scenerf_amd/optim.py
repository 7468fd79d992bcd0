"""AdamW over the renderer's parameters in one HIP launch (``scenerf_hip_adamw_step``, csrc/optim.hip).

The reference trains with ``torch.optim.AdamW(self.parameters(), lr, weight_decay)`` (scenerf.py:756-761).  ``FusedAdamW`` is that
optimizer -- same hyper-parameters, same ``param_groups`` / ``state_dict`` layout (``step``, ``exp_avg``, ``exp_avg_sq`` per parameter),
schedulers work on it unchanged -- with the update of ALL parameters of a group issued as one kernel that reads each gradient where
autograd left it: the renderer hands ``lin_in.weight``'s gradient over as a sliced view of its gradient sink, which the stock fused
optimizer first copies into a contiguous tensor.  amsgrad / maximize are not offered.  CUDA fp32 parameters only: anything else raises
(no fallback).

``capturable=True`` (like torch's): the step count and the learning rate live in a device tensor per parameter group (``hyper`` =
[lr, t, scratch]); the update kernel forms the bias corrections from t + 1 itself and stores t + 1 back when its last workgroup
retires (no increment launch: in a replayed graph that one-element add was a node the whole step hung behind), so the call can sit
inside a captured hipGraph (``scenerf_amd.graph.GraphedStep``) and every replay is one more optimizer step.  A learning rate changed by a
scheduler reaches a captured graph through ``sync_hyper()`` (one small fill, outside the graph).  All parameters of a group step
together in this mode (a parameter without a gradient raises).
"""
from __future__ import annotations

import ctypes as C
from typing import List

import torch

from . import _capi


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 capturable: bool = False):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, capturable=bool(capturable)))
        self._hyper = {}      # index of the group in param_groups -> [device tensor [lr, t, kernel scratch], the lr it holds]

    def _group_index(self, group) -> int:
        for i, g in enumerate(self.param_groups):
            if g is group:
                return i
        raise KeyError("not a parameter group of this optimizer")

    def _group_hyper(self, group, dev):
        gi = self._group_index(group)
        h = self._hyper.get(gi)
        if h is None:
            steps = [float(self.state[p]["step"]) for p in group["params"] if self.state.get(p)]
            t0 = float(max(steps)) if steps else 0.0
            h = [torch.tensor([float(group["lr"]), t0] + [0.0] * _capi.ADAMW_SCRATCH, dtype=torch.float32, device=dev), float(group["lr"])]
            self._hyper[gi] = h
        return h

    def load_state_dict(self, state_dict) -> None:
        """The device-side [lr, t] tensors SURVIVE a load: a hipGraph captured earlier (GraphedStep) holds their addresses, so they are
        refreshed in place -- lr from the loaded group, t = the largest loaded step count -- and the per-parameter ``step`` entries are
        pointed at them again.  (torch's load_state_dict builds new group dicts and new state entries: keyed by the group's INDEX, a
        tensor that was dropped here would leave every later replay counting on freed memory.)"""
        super().load_state_dict(state_dict)
        for gi, group in enumerate(self.param_groups):
            h = self._hyper.get(gi)
            if h is None:
                continue
            steps = [float(self.state[p]["step"]) for p in group["params"] if self.state.get(p) and "step" in self.state[p]]
            h[0][0:1].fill_(float(group["lr"]))
            h[0][1:2].fill_(float(max(steps)) if steps else 0.0)
            h[1] = float(group["lr"])
            if group.get("capturable"):
                for p in group["params"]:
                    if self.state.get(p):
                        self.state[p]["step"] = h[0][1]

    def sync_hyper(self) -> None:
        """Write the groups' current learning rates into their device-side copies (capturable mode; call it after a scheduler step
        and before the next graph replay -- an eager ``step()`` does it by itself)."""
        for gi, group in enumerate(self.param_groups):
            h = self._hyper.get(gi)
            if h is not None and h[1] != float(group["lr"]):
                h[0][0:1].fill_(float(group["lr"]))
                h[1] = float(group["lr"])

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _capi.load()
        for group in self.param_groups:
            entries: List[_capi.AdamWTensor] = []
            keep = []    # (temporaries must outlive the launch call)
            dev = None
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise RuntimeError("FusedAdamW: parameters must be contiguous fp32 CUDA tensors (got %s %s)" % (p.dtype, p.device))
                if g.is_sparse or g.dtype != torch.float32 or g.device != p.device:
                    raise RuntimeError("FusedAdamW: gradients must be dense fp32 tensors on the parameter's device")
                if dev is None:
                    dev = p.device
                elif p.device != dev:
                    raise RuntimeError("FusedAdamW: one device per parameter group")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                if not group.get("capturable"):
                    st["step"] = int(st["step"]) + 1
                e = _capi.AdamWTensor()
                e.p, e.m, e.v = p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                e.numel, e.step = p.numel(), (1 if group.get("capturable") else st["step"])
                if g.is_contiguous():
                    e.g, e.g_cols, e.g_ld = g.data_ptr(), 0, 0
                elif g.dim() == 2 and g.stride(1) == 1 and g.stride(0) >= g.shape[1]:
                    e.g, e.g_cols, e.g_ld = g.data_ptr(), g.shape[1], g.stride(0)     # a column slice of a wider buffer, read in place
                else:
                    gc = g.contiguous()
                    keep.append(gc)
                    e.g, e.g_cols, e.g_ld = gc.data_ptr(), 0, 0
                entries.append(e)
            if not entries:
                continue
            arr = (_capi.AdamWTensor * len(entries))(*entries)
            b1, b2 = group["betas"]
            with torch.cuda.device(dev):
                if group.get("capturable"):
                    if len(entries) != len(group["params"]):
                        raise RuntimeError("FusedAdamW(capturable=True): every parameter of a group needs a gradient (one step count per group)")
                    hyper, lr_held = self._group_hyper(group, dev)
                    if not torch.cuda.is_current_stream_capturing() and lr_held != float(group["lr"]):
                        self.sync_hyper()
                    for p in group["params"]:
                        self.state[p]["step"] = hyper[1]        # (like torch's capturable optimizers: a device scalar; the kernel counts)
                    _capi.check(lib.scenerf_hip_adamw_step_dev(len(entries), arr, hyper.data_ptr(), float(b1), float(b2), float(group["eps"]),
                                                               float(group["weight_decay"]), torch.cuda.current_stream(dev).cuda_stream),
                                "adamw_step_dev")
                else:
                    _capi.check(lib.scenerf_hip_adamw_step(len(entries), arr, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                                           float(group["weight_decay"]), torch.cuda.current_stream(dev).cuda_stream), "adamw_step")
            del keep
        return loss
