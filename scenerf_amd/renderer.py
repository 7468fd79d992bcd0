"""Host side of the MI355X ray renderer: memory, streams and autograd plumbing around the C ABI.

Replaces the body of reference ``SceneRF.batchify_depth_and_color`` (scenerf/models/scenerf.py:598-700)
and its autograd.  All arithmetic happens in libscenerf_hip.so; PyTorch only owns the device buffers,
the current stream and the autograd graph edges:

    PrepareMaps   (C,H,W) fp32 maps  -> (H,W,C) act maps, once per image; its backward hands the (H,W,C)
                  fp32 gradient accumulators back as (C,H,W) grads after *all* chunks have scattered into them
    PackMLP       nn.Linear parameters -> MFMA operand layout, once per call; backward unpacks the grads
    RenderChunk   one chunk of rays: forward = 9 kernel stages, backward = their adjoints

There is no eager/CPU fallback: without the built library every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _capi
from .config import FEAT_CHANNELS, RenderConfig

D_H = _capi.D_HIDDEN
D_L = _capi.D_LATENT
D_X = _capi.D_XENC

# parameter order of one ResnetFC (reference state_dict names, resnetfc.py:88-118)
MLP_PARAM_NAMES: List[str] = ["lin_in.weight", "lin_in.bias", "lin_out.weight", "lin_out.bias"]
for _b in range(3):
    MLP_PARAM_NAMES += ["blocks.%d.fc_0.weight" % _b, "blocks.%d.fc_0.bias" % _b,
                        "blocks.%d.fc_1.weight" % _b, "blocks.%d.fc_1.bias" % _b,
                        "lin_z.%d.weight" % _b, "lin_z.%d.bias" % _b]

OUTPUT_KEYS = ["depth", "color", "gaussian_means", "gaussian_stds", "weights_at_depth", "closest_pts_to_depths",
               "loss_kl", "alphas", "som_vars", "densities", "weights", "depth_volumes"]


def _stream(dev=None) -> int:
    """hipStream_t of torch's current stream ON ``dev`` (the device of the tensors handed to the C ABI, not whatever device
    happens to be current)."""
    return torch.cuda.current_stream(dev).cuda_stream


def _on(dev):
    """Context: make ``dev`` the calling thread's current device for the duration of a group of C-ABI calls.  The library keys its
    per-device state (kernel attributes, descriptor tables) on the current device and launches on the stream it is given, so both
    must be the device the buffers live on -- also when the model sits on a non-current GPU or a process drives several."""
    return torch.cuda.device(dev)


_SIDE = {}
PREFILL_AT = 6               # where a training session zeroes the map-gradient accumulators (420 MB at KITTI: 217 MB for the finest level) on the
                             # side stream: 0 = at the session's start (beside the gaussian head's encode / gather), 1 = behind the head's forward
                             # (beside the sampler, the radiance MLP's encode / gather), 2 = behind the radiance MLP's forward (beside the
                             # per-ray tail and the loss: latency-bound kernels that a 58-us fill over all CUs slows 2-3x: ray_tail_fwd 17 ->
                             # 45 us), 3 = behind the tail's backward (beside lin_out's reduction and the chain), 5 = forked behind the head's
                             # gather, 6 = in the side stream's own order behind the first half of the head's pack (beside the head's
                             # forward: that kernel 108 -> ~150 us, but the sampler, the radiance MLP's encode and its gather run clean:
                             # 14 + 18 + 64 us instead of 17 + 38 + 86).  As replayed hipGraphs, 300-400 steps, three runs each on one box
                             # (r04): 1: 2.569-2.583 ms, 2: 2.587-2.623, 3: 2.627-2.632; later in the round 1: 2.499-2.513, 6 with
                             # FILL_WORKGROUPS 64 / 128 / 256: 2.471-2.492 / 2.474-2.488 / 2.476-2.489, 6 with torch's fill 2.515-2.528,
                             # 5 with 64 workgroups 2.518-2.524.  (r03, eager issue: 0: 2.635, 2: 2.622.)
FILL_WORKGROUPS = 128        # > 0: that fill by the library's bounded streaming-store kernel (scenerf_hip_fill_zero) with this many workgroups
                             # instead of torch's fill kernel
CHAIN_FIRST = True           # capture / issue order: at every fork the critical chain's next kernel is launched BEFORE the side stream's, and
                             # the packs are launched behind the chain's first kernels (below: "Launch order")
MAIN_WGRAD_OVERLAP = True    # radiance MLP's weight gradients on the library's side stream, beside its feature-map gradients (-35 us/step)
DEFER_HEAD_PACK = True     # (tools/ab_step.py toggles this)
GRADS_BEHIND_HEAD = True   # the radiance MLP's weight / feature-map gradients start only when the gaussian head's backward kernels are done
MAP_GRADS_ON_SIDE = True   # ... and its feature-map gradients run on the side stream, unjoined until the accumulators' next consumer
# ... also in sessions of several chunks (the trainer's S source frames per image, RenderChunk._backward): measured on the trainer's step,
# same box, three alternations -- replayed 7.59 / 7.58 / 7.60 ms against 7.23 / 7.23 / 7.37 without, eager equal (7.46 vs 7.47): the next
# chunk's head backward queues behind the previous chunk's scatter on the side stream and the join holds its weight gradients back.  Off.
MAP_GRADS_ON_SIDE_MULTI = False
SPLIT_HEAD_PACK = True     # the head's pack in two calls, its forward's operands first



def reset_side_streams() -> None:
    """Forget the renderer's side streams (new ones are made on demand).  For whoever aborted a stream capture half-way: the side streams
    that were forked into it stay unusable on ROCm ("invalid argument" at their next launch) -- GraphedStep calls this when a capture fails."""
    _SIDE.clear()


def _side_stream(dev) -> "torch.cuda.Stream":
    key = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=key)
    return _SIDE[key]


_CONSTS = {}


def _sampler_constants(dev, U: int, G: int, max_depth: float):
    """The two linspace vectors of the samplers (utils.py:79-81, scenerf.py:556-560): functions of the configuration only, so they are
    built once per (device, configuration) -- on the host, copied synchronously -- instead of with two launches per chunk."""
    key = (torch.device(dev).index, U, G, max_depth)
    hit = _CONSTS.get(key)
    if hit is None:
        lin_u = torch.linspace(0.2, max_depth, steps=U, dtype=torch.float32).to(dev) if U > 0 else None
        step = max_depth * 1.0 / G
        anchors = torch.linspace(step / 2, max_depth - step / 2, steps=G, dtype=torch.float32).to(dev)
        hit = _CONSTS[key] = (lin_u, anchors)
    return hit


_ZERO1 = {}


def _zero_token(dev) -> torch.Tensor:
    """The gradient RenderChunk.backward returns for its three autograd tokens: a zero nobody reads (PackMLP / PrepareMaps ignore the
    value), created once per device instead of by a fill launch per backward."""
    key = torch.device(dev).index
    z = _ZERO1.get(key)
    if z is None:
        z = _ZERO1[key] = torch.zeros(1, dtype=torch.float32, device=dev)
    return z


def _require_cuda(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU (got %s): the SceneRF hot path has no CPU fallback" % (name, t.device))


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype is torch.float32 and t.is_contiguous():     # (the common case: `.to` alone is ~9 us of dispatch, ~65 calls per step)
        return t.detach() if t.requires_grad else t
    return t.detach().to(torch.float32).contiguous()


def _act_dtype(prec: int):
    return torch.bfloat16 if prec else torch.float32


# ------------------------------------------------------------------------------------------------ feature maps
class HWC:
    """Marks a feature map that is handed to ``render_rays_batch`` as an fp32 ``(H, W, C)`` tensor instead of the reference's
    ``(C, H, W)``: ``x_rgb["1_1"] = HWC(t)``.  The renderer then reads it in place and returns its gradient in ``(H, W, C)`` -- neither
    the per-call ``(C,H,W) -> (H,W,C)`` copy nor the transpose of the gradient accumulator happens (0.31 ms per KITTI step).  The entry
    for a producer that emits channels-last maps (``SphereResampler(layout="hwc")``); the stock encoder's ``(C, H, W)`` maps keep working
    unwrapped, and the two can be mixed per pyramid level."""
    __slots__ = ("t",)

    def __init__(self, t: torch.Tensor):
        if t.dim() != 3:
            raise ValueError("HWC expects an (H, W, C) tensor, got shape %s" % (tuple(t.shape),))
        self.t = t

    device = property(lambda self: self.t.device)
    shape = property(lambda self: self.t.shape)
    _version = property(lambda self: self.t._version)

    def data_ptr(self):
        return self.t.data_ptr()

    def detach(self) -> "HWC":
        return HWC(self.t.detach())


def draw_noise_g(cfg: RenderConfig, R: int, device) -> torch.Tensor:
    """N(0,1) noise of the probabilistic sampler, (R, G*P).  utils.py:208-211 draws it on the CPU generator: same stream here
    (bit-identical to torch.normal(zeros, ones)), into pinned memory so that the H2D copy does not stall the host; with
    ``cfg.device_rng`` on the device generator instead."""
    GP = cfg.n_gaussians * cfg.n_pts_per_gaussian
    if cfg.device_rng:
        return torch.randn((R, GP), dtype=torch.float32, device=device)
    return torch.empty((R, GP), dtype=torch.float32, pin_memory=True).normal_().to(device, non_blocking=True)


class MapHolder:
    """(H,W,C) copies of the 5 encoder maps + lazily allocated fp32 gradient accumulators."""

    def __init__(self, cfg: RenderConfig):
        self.cfg = cfg
        self.hwc: List[torch.Tensor] = []
        self.shapes = []
        self.gmaps: Optional[List[torch.Tensor]] = None
        self._gflat: Optional[torch.Tensor] = None      # the one allocation the five accumulators are views of
        self.debug_aux: Optional[Dict[str, torch.Tensor]] = None   # dict when the session was opened with debug_aux=True
        self.rng: Optional[torch.Tensor] = None     # device int64 [3] {seed, calls, scratch}: in-kernel sampler noise (device_rng sessions)
        # caller-owned dict (the model's): the converted (H,W,C) copies of the last image's maps, keyed by the source tensors' identity and
        # version counters -- the S source frames of one image (scenerf.py:154-156: x_rgb is made once per image, rendered from S poses, plus
        # the metric-only renders) then convert once instead of 2 S times (0.15 ms of 420 MB layout conversion per call)
        self.convert_cache: Optional[dict] = None

    def convert(self, chw: Sequence[torch.Tensor]) -> None:
        """Also serves as the refresh of a long-lived session (inference.ImageRenderer): a second call writes into the SAME converted
        buffers, so device addresses captured in a hipGraph stay valid while the contents follow the caller's maps."""
        lib = _capi.load()
        prec = self.cfg.precision_code
        want = self.cfg.map_shapes()
        old = self.hwc if len(self.hwc) == len(chw) else None
        self.hwc, self.shapes = [], []
        for i, t in enumerate(chw):
            _require_cuda(t, "x_rgb map %d" % i)
            if i not in self.cfg.hwc_scales and tuple(t.shape) != tuple(want[i]):
                raise RuntimeError("feature map %d has shape %s, expected %s for sphere %dx%d" % (
                    i, tuple(t.shape), want[i], self.cfg.sphere_W, self.cfg.sphere_H))
            if i in self.cfg.hwc_scales:   # fp32 (H,W,C), read in place (see HWC)
                _require_cuda(t, "x_rgb map %d" % i)
                c, h, w = want[i]
                if tuple(t.shape) != (h, w, c):
                    raise RuntimeError("feature map %d (HWC) has shape %s, expected %s" % (i, tuple(t.shape), (h, w, c)))
                self.hwc.append(self._in_place(_f32c(t), old, i))
                self.shapes.append((c, h, w))
                continue
            src = _f32c(t)
            c, h, w = src.shape
            if i in self.cfg.direct_scales:
                dst = self._in_place(src, old, i)   # read in place, (C,H,W) fp32 (RenderConfig.direct_scales)
            else:
                hit = None
                # (keyed on the caller's tensor; `src` is it, or its detached alias -- a converted temporary has no identity to key on)
                cacheable = old is None and self.convert_cache is not None and src.data_ptr() == t.data_ptr() and t.dtype is torch.float32
                if cacheable:
                    # the capture this stream records into (0: none).  A buffer converted inside a capture is graph memory that only
                    # holds the maps during a replay: its entry is reused by the later render calls of the SAME capture (the trainer's
                    # per-image step under graph.GraphedFn) and by nothing else
                    cid = C.c_ulonglong(0)
                    _capi.check(lib.scenerf_hip_stream_capture_id(_stream(src.device), C.byref(cid)), "stream_capture_id")
                    ckey = (prec, cid.value)
                    ent = self.convert_cache.get(i)
                    if ent is not None and ent[0] is t and ent[1] == t._version and ent[2] == t.data_ptr() and ent[3] == ckey \
                            and tuple(ent[4].shape) == (h, w, c):
                        hit = ent[4]
                if hit is not None:
                    # (the conversion that made it ran on a stream the current one has been ordered behind: both are this device's
                    #  current stream of the same caller thread, or the entry's event says so)
                    torch.cuda.current_stream(src.device).wait_event(ent[5])
                    dst = hit
                else:
                    dst = old[i] if old is not None else torch.empty((h, w, c), dtype=_act_dtype(prec), device=src.device)
                    _capi.check(lib.scenerf_hip_maps_chw_to_hwc(src.data_ptr(), dst.data_ptr(), c, h, w, prec, _stream(src.device)),
                                "maps_chw_to_hwc")
                    if cacheable:
                        self.convert_cache[i] = (t, t._version, t.data_ptr(), ckey, dst, torch.cuda.current_stream(src.device).record_event())
            self.hwc.append(dst)
            self.shapes.append((c, h, w))

    @staticmethod
    def _in_place(src: torch.Tensor, old, i: int) -> torch.Tensor:
        """A level that is read where the caller keeps it.  On a refresh the address already handed out must stay the one that is
        read: same memory -> nothing to do; a temporary (the caller's map was not fp32-contiguous) or a moved map -> copy into the
        buffer of the first call."""
        if old is None or old[i].data_ptr() == src.data_ptr():
            return src
        old[i].copy_(src)
        return old[i]

    def _alloc_gmaps(self, zero_now: bool) -> None:
        dev = self.hwc[0].device
        # (H,W,C) accumulators, transposed once at the end; the direct scales accumulate in the (C,H,W) result itself.  One flat
        # allocation carved into the five maps: ONE fill launch zeroes them all (five tensors cost five to six multi-tensor launches)
        shapes = [(c, h, w) if i in self.cfg.direct_scales else (h, w, c) for i, (c, h, w) in enumerate(self.shapes)]
        sizes = [a * b * c for a, b, c in shapes]
        pad = [(n + 63) // 64 * 64 for n in sizes]          # 256-byte aligned starts
        self._gflat = torch.empty(sum(pad), dtype=torch.float32, device=dev)
        self.gmaps, off = [], 0
        for shp, n, p in zip(shapes, sizes, pad):
            self.gmaps.append(self._gflat[off:off + n].view(*shp))
            off += p
        if zero_now:
            self._gflat.zero_()

    def prefill_grad_accumulators(self, after=None) -> None:
        """Called from the forward when a map gradient will be asked for: the accumulators (420 MB at the KITTI shapes) are zeroed on
        the side stream, under the forward's MFMA-bound kernels, instead of on the backward's critical path (81 us of fills, r02_d).
        ``after``: an event of the current stream the fill waits for (instead of everything the stream holds now) -- recorded where
        the buffer was allocated (``alloc_grad_accumulators``), so that the caller can launch the chain's next kernel first."""
        if self.gmaps is None:
            self._alloc_gmaps(False)
        elif after is None:
            return
        if getattr(self, "_gmaps_filled", False):
            return
        self._gmaps_filled = True
        dev = self.hwc[0].device
        main, side = torch.cuda.current_stream(dev), _side_stream(dev)
        if after is not None:
            side.wait_event(after)
        else:
            side.wait_stream(main)    # the blocks may have just been freed by work still queued on the main stream
        with torch.cuda.stream(side):
            if FILL_WORKGROUPS > 0 and self._gflat.numel() % 4 == 0:
                with torch.cuda.device(dev):
                    _capi.check(_capi.load().scenerf_hip_fill_zero(self._gflat.data_ptr(), self._gflat.numel() * 4, FILL_WORKGROUPS,
                                                                   side.cuda_stream), "fill_zero")
            else:
                self._gflat.zero_()
            self._gmaps_ready = side.record_event()
        # if no backward ever waits on the event (graph dropped, exception): the allocator must not hand this block to a main-stream
        # tenant while the side-stream fill is still pending
        self._gflat.record_stream(side)

    def alloc_grad_accumulators(self):
        """Allocate the accumulators now (current stream) and return the event a later ``prefill_grad_accumulators(after=...)`` orders
        its fill behind: whatever used the blocks before is in front of it."""
        if self.gmaps is not None:
            return None
        self._alloc_gmaps(False)
        self._gmaps_filled = False
        return torch.cuda.current_stream(self.hwc[0].device).record_event()

    def join_prefill(self) -> None:
        """Order the current stream behind the side-stream fill NOW (the backward's ``grad_accumulators()`` then has nothing to wait
        for): called where the stream joins the side stream anyway, so that the two waits are one node of a captured graph."""
        ev, self._gmaps_ready = getattr(self, "_gmaps_ready", None), None
        if ev is not None:
            torch.cuda.current_stream(self.hwc[0].device).wait_event(ev)

    def grad_accumulators(self, wait: bool = True) -> List[torch.Tensor]:
        """``wait=False``: only the buffers (their addresses) -- for a caller that launches its scatter on the stream the previous one ran
        on (the side stream: in order behind it) and must not pull the current stream behind the previous chunk's scatter."""
        if self.gmaps is None:
            self._alloc_gmaps(True)
        if wait:
            ev, self._gmaps_ready = getattr(self, "_gmaps_ready", None), None
            cur = torch.cuda.current_stream(self.gmaps[0].device)
            # (recorded on the side stream; a consumer ON the side stream is behind it already -- and a stream waiting for its own event
            #  inside a capture ends in a segfault of hipStreamEndCapture, ROCm 7.2: the gaussian head's backward of the NEXT chunk)
            if ev is not None and cur != _side_stream(self.gmaps[0].device):
                cur.wait_event(ev)
            elif ev is not None:
                self._gmaps_ready = ev       # still what the next consumer on another stream has to wait for
        return self.gmaps

    def map_ptr_array(self):
        return (C.c_void_p * 5)(*[t.data_ptr() for t in self.hwc])

    def gmap_ptr_array(self, wait: bool = True):
        return (C.c_void_p * 5)(*[t.data_ptr() for t in self.grad_accumulators(wait)])


class PrepareMaps(torch.autograd.Function):
    @staticmethod
    def forward(ctx, holder: MapHolder, *chw):
        holder.convert(chw)
        ctx.holder = holder
        # (needs_input_grad is the tensors' requires_grad whatever the grad mode: a metric-only render under torch.no_grad() of maps that
        #  are trained elsewhere in the step must not zero 420 MB of accumulators nobody will read)
        if any(ctx.needs_input_grad[1:]) and getattr(holder, "grad_mode", True):
            if PREFILL_AT == 0:
                holder.prefill_grad_accumulators()
            else:
                holder._want_prefill = True     # (zeroed from the first chunk's forward: RenderChunk._forward)
                if PREFILL_AT == 6:
                    # allocated HERE, with an event of this point of the stream: the fill is queued later, on the side stream, behind
                    # this event only -- whatever used the block before is in front of it, and nothing freed later can be handed out
                    holder._fill_after = holder.alloc_grad_accumulators()
        return torch.empty(1, device=chw[0].device)   # autograd token: its value is never read (no fill launch)

    @staticmethod
    def backward(ctx, _g):
        holder: MapHolder = ctx.holder
        if holder.gmaps is None:
            return (None,) + tuple(None for _ in holder.shapes)
        lib = _capi.load()
        outs = []
        holder.grad_accumulators()   # (orders this stream after the zero fills if no chunk's backward has done so)
        with _on(holder.gmaps[0].device):
            outs = PrepareMaps._transpose_back(ctx, holder, lib)
        holder.gmaps = holder._gflat = None
        return (None,) + tuple(outs)

    @staticmethod
    def _transpose_back(ctx, holder, lib):
        outs = []
        for i, (c, h, w) in enumerate(holder.shapes):
            if not ctx.needs_input_grad[1 + i]:
                outs.append(None)
                continue
            if i in holder.cfg.direct_scales or i in holder.cfg.hwc_scales:   # the accumulator already has the input's layout
                outs.append(holder.gmaps[i])
                continue
            g = torch.empty((c, h, w), dtype=torch.float32, device=holder.gmaps[i].device)
            _capi.check(lib.scenerf_hip_grads_hwc_to_chw(holder.gmaps[i].data_ptr(), g.data_ptr(), c, h, w, _stream(g.device)),
                        "grads_hwc_to_chw")
            outs.append(g)
        return outs


# ------------------------------------------------------------------------------------------------ MLP operands
class PackedMLP:
    """ResnetFC parameters in the operand layout of include/scenerf_hip.h::scenerf_mlp_weights."""

    def __init__(self, params: Sequence[torch.Tensor], d_out: int, cfg: RenderConfig, pack_stream=None, defer: bool = False,
                 split: bool = False):
        """``pack_stream``: launch the pack there instead of on the current stream.  ``defer`` (with a pack stream): do not launch yet --
        ``launch_pack()`` (or the first ``wait_ready()``) does, ordered behind the point of the current stream this constructor ran at
        (where the parameters were last written), not behind whatever the current stream has been given since.  ``split`` (with a pack
        stream, bf16): the pack in two calls, a forward's operands first (SCENERF_FLAG_PACK_FORWARD / _REST) -- ``wait_ready()`` then
        waits for the first, ``wait_ready(backward=True)`` for both."""
        p = dict(zip(MLP_PARAM_NAMES, [_f32c(t) for t in params]))
        for n, t in p.items():
            _require_cuda(t, n)
        prec = cfg.precision_code
        act = _act_dtype(prec)
        dev = p["lin_in.weight"].device
        if tuple(p["lin_out.weight"].shape) != (d_out, D_H) or tuple(p["lin_in.weight"].shape) != (D_H, 42):
            raise RuntimeError("ResnetFC parameter shapes do not match d_in=42, d_hidden=512, d_out=%d" % d_out)
        self.d_out = d_out
        self.prec = prec
        self.device = dev
        self.params = p   # keep the fp32 parameters alive: several operand pointers alias them
        # one act-typed and one fp32 buffer hold every packed operand; scenerf_hip_mlp_pack fills them in two launches
        first_cols = (3 * D_X + D_L) if prec else D_L
        act_sizes = [("w_h0", D_H * first_cols), ("w_h1", D_H * (D_H + D_L)), ("w_h2", D_H * (D_H + D_L)), ("w_h3", D_H * D_H)]
        for b in range(3):
            act_sizes += [("w_fc0.%d" % b, D_H * D_H), ("w_fc0_t.%d" % b, D_H * D_H), ("w_fc1_t.%d" % b, D_H * D_H)]
        for i, c in enumerate(FEAT_CHANNELS):
            act_sizes.append(("w_z_t.%d" % i, c * 3 * D_H))
        if prec:   # forward operands re-tiled in 16 KiB streaming blocks for the fused trunk kernel (scenerf_hip.h)
            act_sizes.append(("w_stream", _capi.W_STREAM_BLOCKS * 8192))
        tot = sum(((n + 7) // 8) * 8 for _, n in act_sizes)
        self.act_buf = torch.empty(tot, dtype=act, device=dev)
        av, off = {}, 0
        for name, n in act_sizes:
            av[name] = self.act_buf[off:off + n]
            off += ((n + 7) // 8) * 8
        self.f32_buf = torch.empty(D_H * D_X + 4 * D_H, dtype=torch.float32, device=dev)
        w_in = self.f32_buf[:D_H * D_X]
        b_h = [self.f32_buf[D_H * D_X + i * D_H: D_H * D_X + (i + 1) * D_H] for i in range(4)]
        s = _capi.MlpWeights()
        s.d_out = d_out
        s.w_in, s.b_in = w_in.data_ptr(), p["lin_in.bias"].data_ptr()
        for i in range(4):
            s.w_h[i], s.b_h[i] = av["w_h%d" % i].data_ptr(), b_h[i].data_ptr()
        for i in range(3):
            s.w_fc0[i], s.b_fc0[i] = av["w_fc0.%d" % i].data_ptr(), p["blocks.%d.fc_0.bias" % i].data_ptr()
            s.w_fc0_t[i], s.w_fc1_t[i] = av["w_fc0_t.%d" % i].data_ptr(), av["w_fc1_t.%d" % i].data_ptr()
        s.w_out, s.b_out = p["lin_out.weight"].data_ptr(), p["lin_out.bias"].data_ptr()
        for i in range(5):
            s.w_z_t[i] = av["w_z_t.%d" % i].data_ptr()
        s.w_stream = av["w_stream"].data_ptr() if prec else None
        # a training session's pack launch also zeroes the gradient sink of the backward to come (no fill launch of its own: on a
        # replayed step every extra node is a queue the executor may serialise the critical chain behind -- r04_h)
        self.gflat: Optional[torch.Tensor] = None
        if pack_stream is not None:
            self.gflat = torch.empty(self._sink_numel(), dtype=torch.float32, device=dev)
            s.clear, s.clear_floats = self.gflat.data_ptr(), self.gflat.numel()
        else:
            s.clear, s.clear_floats = None, 0
        self.c = s
        raw = _capi.MlpParams()
        raw.d_out = d_out
        raw.lin_in_w, raw.lin_in_b = p["lin_in.weight"].data_ptr(), p["lin_in.bias"].data_ptr()
        raw.lin_out_w, raw.lin_out_b = p["lin_out.weight"].data_ptr(), p["lin_out.bias"].data_ptr()
        for i in range(3):
            raw.fc0_w[i], raw.fc0_b[i] = p["blocks.%d.fc_0.weight" % i].data_ptr(), p["blocks.%d.fc_0.bias" % i].data_ptr()
            raw.fc1_w[i], raw.fc1_b[i] = p["blocks.%d.fc_1.weight" % i].data_ptr(), p["blocks.%d.fc_1.bias" % i].data_ptr()
            raw.linz_w[i], raw.linz_b[i] = p["lin_z.%d.weight" % i].data_ptr(), p["lin_z.%d.bias" % i].data_ptr()
        self._raw = raw
        ccfg = cfg.to_c()
        # pack_stream: launch the pack there instead of on the current stream (the radiance MLP's operands are first read ~0.3 ms into
        # a training step, after the gaussian head's chain: its pack runs beside that chain); wait_ready() orders the consumer
        self._ready = None
        self._ready_rest = None
        self.params_written = None       # event of the stream position the parameters were last written at (deferred packs)
        self._pack_stream = pack_stream
        self._split = bool(split and pack_stream is not None and prec == 1)
        self._pending = None
        self._packed_ev = None           # event behind the complete pack (set by launch_pack / below)
        if pack_stream is not None:
            # the parameters were last written on the current stream: an event HERE, so that a deferred launch does not also wait for
            # the kernels the current stream is given in between
            self.params_written = torch.cuda.current_stream(dev).record_event()
            self._pending = (ccfg, pack_stream, self.params_written, 0)
            if not defer:
                self.launch_pack()
        else:
            _capi.check(_capi.load().scenerf_hip_mlp_pack(C.byref(ccfg), C.byref(raw), C.byref(s), _stream(dev)), "mlp_pack")
        # gradient sink (flat fp32 buffer carved into the scenerf_mlp_grads fields): allocated and zeroed on first backward, or zeroed by
        # the pack launch above
        self.gviews: Dict[str, torch.Tensor] = {}
        self.gc: Optional[_capi.MlpGrads] = None

    _GRAD_FIELDS = None

    def same_storage(self, params: Sequence[torch.Tensor]) -> bool:
        """Whether ``params`` are still the tensors (addresses) this object's operand pointers alias."""
        return all(_f32c(t).data_ptr() == self.params[n].data_ptr() for n, t in zip(MLP_PARAM_NAMES, params))

    def repack(self, cfg: RenderConfig) -> None:
        """Run the pack again into the same operand buffers (current stream): the packed operands follow parameter VALUES that
        changed in place since construction, whatever way they were written (optimizer step, ``p.data.copy_``, ``load_state_dict``),
        while every device address a captured hipGraph holds stays valid.  Requires ``same_storage``."""
        self.wait_ready(backward=True)
        self.c.clear, self.c.clear_floats = None, 0     # (a sink zeroed by the first pack may hold gradients by now)
        _capi.check(_capi.load().scenerf_hip_mlp_pack(C.byref(cfg.to_c()), C.byref(self._raw), C.byref(self.c), _stream(self.device)), "mlp_pack")

    def launch_pack(self, upto: int = 2) -> None:
        """Launch a deferred pack on its stream (no-op if there is none pending).  ``upto`` = 1: only the first half of a split pack
        (the caller has something to put between the halves on the pack stream)."""
        pend = self._pending
        if pend is None:
            return
        ccfg, pack_stream, params_written, done = pend
        lib = _capi.load()
        if done == 0:
            pack_stream.wait_event(params_written)
        with torch.cuda.device(self.device):
            if self._split:
                cf = type(ccfg).from_buffer_copy(ccfg)
                if done == 0:
                    cf.flags |= _capi.FLAG_PACK_FORWARD
                    _capi.check(lib.scenerf_hip_mlp_pack(C.byref(cf), C.byref(self._raw), C.byref(self.c), pack_stream.cuda_stream), "mlp_pack")
                    self._ready = pack_stream.record_event()
                    done = 1
                if upto >= 2:
                    cf.flags = (cf.flags & ~_capi.FLAG_PACK_FORWARD) | _capi.FLAG_PACK_REST
                    _capi.check(lib.scenerf_hip_mlp_pack(C.byref(cf), C.byref(self._raw), C.byref(self.c), pack_stream.cuda_stream), "mlp_pack")
                    self._ready_rest = pack_stream.record_event()
                    done = 2
            else:
                _capi.check(lib.scenerf_hip_mlp_pack(C.byref(ccfg), C.byref(self._raw), C.byref(self.c), pack_stream.cuda_stream), "mlp_pack")
                self._ready = pack_stream.record_event()
                done = 2
        if done < 2:
            self._pending = (ccfg, pack_stream, params_written, done)
            return
        self._pending = None
        self._packed_ev = self._ready_rest if self._ready_rest is not None else self._ready    # (kept: a later session that reuses the operands)
        # if no consumer ever waits on the event (exception, graph dropped): the allocator must not hand these blocks to a tenant of
        # another stream while the pack is still pending
        for t in (self.gflat, self.act_buf, self.f32_buf):
            if t is not None:
                t.record_stream(pack_stream)

    def wait_ready(self, backward: bool = False) -> None:
        """Order the current stream after a pack that was launched on another stream (no-op otherwise, and after the first call).
        ``backward``: also after the second half of a split pack (transposed operands, W_z^T, the zeroed gradient sink)."""
        self.launch_pack()
        if backward and self._ready_rest is not None:     # (the second half is behind the first on the pack stream)
            ev, self._ready_rest, self._ready = self._ready_rest, None, None
            cur = torch.cuda.current_stream(self.device)
            if cur != self._pack_stream:                   # (the pack stream itself is behind its own launches)
                cur.wait_event(ev)
            return
        ev, self._ready = self._ready, None
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)

    def _sink_fields(self):
        d = self.d_out
        fields = [("w_in", (D_H, _capi.WIN_LD)), ("b_in", (D_H,))]   # (columns >= 48 of w_in: scratch of the batched wgrad kernel)
        for b in range(3):
            fields += [("w_fc0.%d" % b, (D_H, D_H)), ("b_fc0.%d" % b, (D_H,)),
                       ("w_fc1.%d" % b, (D_H, D_H)), ("b_fc1.%d" % b, (D_H,))]
        fields += [("w_z", (3 * D_H, D_L)), ("b_z", (3 * D_H,)), ("w_out", (d, D_H)), ("b_out", (d,)), ("w_in_dense", (D_H, 42))]
        return fields

    def _sink_numel(self) -> int:
        n = 0
        for _, shp in self._sink_fields():
            k = 1
            for x in shp:
                k *= x
            n += k
        return n

    def grad_sink(self) -> "_capi.MlpGrads":
        if self.gc is not None:
            return self.gc
        fields = self._sink_fields()
        if self.gflat is None:
            self.gflat = torch.zeros(self._sink_numel(), dtype=torch.float32, device=self.device)
        off = 0
        for name, shp in fields:
            n = 1
            for x in shp:
                n *= x
            self.gviews[name] = self.gflat[off:off + n].view(*shp)
            off += n
        g = _capi.MlpGrads()
        g.w_in, g.b_in = self.gviews["w_in"].data_ptr(), self.gviews["b_in"].data_ptr()
        for b in range(3):
            g.w_fc0[b], g.b_fc0[b] = self.gviews["w_fc0.%d" % b].data_ptr(), self.gviews["b_fc0.%d" % b].data_ptr()
            g.w_fc1[b], g.b_fc1[b] = self.gviews["w_fc1.%d" % b].data_ptr(), self.gviews["b_fc1.%d" % b].data_ptr()
        g.w_z, g.b_z = self.gviews["w_z"].data_ptr(), self.gviews["b_z"].data_ptr()
        g.w_out, g.b_out = self.gviews["w_out"].data_ptr(), self.gviews["b_out"].data_ptr()
        g.w_in_dense = self.gviews["w_in_dense"].data_ptr()
        self.gc = g
        return g

    def unpack_grads(self) -> List[Optional[torch.Tensor]]:
        """Gradients in MLP_PARAM_NAMES order (None if no chunk ran a backward)."""
        if self.gflat is None or not self.gviews:
            return [None] * len(MLP_PARAM_NAMES)
        v = self.gviews
        # (lin_in.weight: the dense copy scenerf_hip_mlp_backward leaves next to the 256-wide sink -- a column slice handed to autograd
        # was cloned by AccumulateGrad, one launch per MLP in front of the optimizer)
        out = {"lin_in.weight": v["w_in_dense"], "lin_in.bias": v["b_in"],
               "lin_out.weight": v["w_out"], "lin_out.bias": v["b_out"]}
        for b in range(3):
            out["blocks.%d.fc_0.weight" % b] = v["w_fc0.%d" % b]
            out["blocks.%d.fc_0.bias" % b] = v["b_fc0.%d" % b]
            out["blocks.%d.fc_1.weight" % b] = v["w_fc1.%d" % b]
            out["blocks.%d.fc_1.bias" % b] = v["b_fc1.%d" % b]
            out["lin_z.%d.weight" % b] = v["w_z"][b * D_H:(b + 1) * D_H]
            out["lin_z.%d.bias" % b] = v["b_z"][b * D_H:(b + 1) * D_H]
        return [out[n] for n in MLP_PARAM_NAMES]


class GenericPackedMLP:
    """A ResnetFC of a shape the fused kernels are not built for (anything but 3 blocks x 512; ``ResnetFC.ordered_params`` order): the
    reference's parameter layout, lin_in's input columns zero-padded 42 -> 48 and lin_out's rows to a multiple of 8, for
    ``scenerf_hip_resnetfc_forward`` / ``_forward_train`` / ``_backward`` (fp32, one MFMA GEMM per nn.Linear and per gradient: round 6 made
    this path trainable).  Same interface as ``PackedMLP`` where the chunk touches it."""
    generic = True

    def __init__(self, params: Sequence[torch.Tensor], d_out: int, cfg: RenderConfig, trainable: bool = False):
        if cfg.precision_code != 0:
            raise RuntimeError("a ResnetFC other than 3 blocks x 512 runs on the per-layer fp32 GEMM path only: construct the model with "
                               "precision='fp32' (the bf16 kernels are built for the trunk SceneRF instantiates)")
        ps = [_f32c(t.detach()) for t in params]
        if len(ps) < 10 or (len(ps) - 4) % 6:
            raise RuntimeError("GenericPackedMLP: expected lin_in, lin_out and six tensors per block (ResnetFC.ordered_params)")
        for i, t in enumerate(ps):
            _require_cuda(t, "ResnetFC parameter %d" % i)
        nb = (len(ps) - 4) // 6
        H = ps[0].shape[0]
        if tuple(ps[0].shape) != (H, 42) or tuple(ps[2].shape) != (d_out, H) or nb > _capi.RESNETFC_MAX_BLOCKS or H % 16:
            raise RuntimeError("ResnetFC parameter shapes do not match d_in=42, d_hidden=%d, d_out=%d" % (H, d_out))
        self.d_out, self.d_hidden, self.n_blocks, self.device = d_out, H, nb, ps[0].device
        self.d_out_pad = (d_out + 7) // 8 * 8
        dev = self.device
        w_in = torch.zeros((H, D_X), dtype=torch.float32, device=dev)
        w_in[:, :42] = ps[0]
        w_out = torch.zeros((self.d_out_pad, H), dtype=torch.float32, device=dev)
        w_out[:d_out] = ps[2]
        b_out = torch.zeros((self.d_out_pad,), dtype=torch.float32, device=dev)
        b_out[:d_out] = ps[3]
        self._keep = ps + [w_in, w_out, b_out]        # (the struct below holds raw pointers)
        n = _capi.ResnetFCNet()
        n.n_blocks, n.d_hidden, n.d_out_pad = nb, H, self.d_out_pad
        n.w_in, n.b_in, n.w_out, n.b_out = w_in.data_ptr(), ps[1].data_ptr(), w_out.data_ptr(), b_out.data_ptr()
        for b in range(nb):
            w0, b0, w1, b1, wz, bz = ps[4 + 6 * b: 10 + 6 * b]
            if tuple(w0.shape) != (H, H) or tuple(w1.shape) != (H, H) or tuple(wz.shape) != (H, D_L):
                raise RuntimeError("ResnetFC block %d: parameter shapes do not match d_hidden=%d, d_latent=%d" % (b, H, D_L))
            n.w_fc0[b], n.b_fc0[b], n.w_fc1[b], n.b_fc1[b], n.w_z[b], n.b_z[b] = (w0.data_ptr(), b0.data_ptr(), w1.data_ptr(), b1.data_ptr(),
                                                                                 wz.data_ptr(), bz.data_ptr())
        self.c = n
        self.trainable = trainable
        self.gflat, self.gc, self.gviews = None, None, {}
        self.ct = None
        if trainable:
            # transposed operands of the input-gradient GEMMs (scenerf_resnetfc_t): nn.Linear.weight^T per fc layer, lin_out.weight^T padded to 16
            # output columns, and per pyramid level the columns of [Wz_0; ..; Wz_{nb-1}] as rows: [C_s][nb * H]
            t = _capi.ResnetFCNetT()
            keep = []
            for b in range(nb):
                w0t, w1t = ps[4 + 6 * b].t().contiguous(), ps[6 + 6 * b].t().contiguous()
                keep += [w0t, w1t]
                t.w_fc0_t[b], t.w_fc1_t[b] = w0t.data_ptr(), w1t.data_ptr()
            w_out_t = torch.zeros((H, 16), dtype=torch.float32, device=dev)
            w_out_t[:, :d_out] = ps[2].t()
            t.w_out_t = w_out_t.data_ptr()
            wz_all = torch.cat([ps[8 + 6 * b] for b in range(nb)], dim=0)          # [nb * H][2480]
            off = 0
            for sc, (c, _, _) in enumerate(cfg.map_shapes()):
                wzt = wz_all[:, off:off + c].t().contiguous()                       # [C_s][nb * H]
                keep.append(wzt)
                t.w_z_t[sc] = wzt.data_ptr()
                off += c
            self._keep += keep + [w_out_t]
            self.ct = t

    def launch_pack(self, upto: int = 2) -> None:
        pass

    def wait_ready(self, backward: bool = False) -> None:
        if backward and not self.trainable:
            raise RuntimeError("this ResnetFC session was opened without gradients (a ResnetFC other than 3 blocks x 512 trains in fp32 "
                               "when its parameters require them)")

    # ---- gradient sink: one flat fp32 buffer (one all-reduce per MLP, like PackedMLP) carved into scenerf_resnetfc_grads
    def _sink_fields(self):
        H, nb = self.d_hidden, self.n_blocks
        f = [("w_in", (H, D_X)), ("b_in", (H,)), ("w_z", (nb * H, D_L)), ("w_out", (16, H)), ("b_out", (16,))]
        for b in range(nb):
            f += [("w_fc0.%d" % b, (H, H)), ("b_fc0.%d" % b, (H,)), ("w_fc1.%d" % b, (H, H)), ("b_fc1.%d" % b, (H,))]
        return f

    def grad_sink(self) -> "_capi.ResnetFCGrads":
        if self.gc is not None:
            return self.gc
        fields = self._sink_fields()
        if self.gflat is None:
            self.gflat = torch.zeros(sum(int(np.prod(shp)) for _, shp in fields), dtype=torch.float32, device=self.device)
        off = 0
        for name, shp in fields:
            n = int(np.prod(shp))
            self.gviews[name] = self.gflat[off:off + n].view(*shp)
            off += n
        g = _capi.ResnetFCGrads()
        v = self.gviews
        g.w_in, g.b_in, g.w_z, g.w_out, g.b_out = (v["w_in"].data_ptr(), v["b_in"].data_ptr(), v["w_z"].data_ptr(), v["w_out"].data_ptr(),
                                                  v["b_out"].data_ptr())
        for b in range(self.n_blocks):
            g.w_fc0[b], g.b_fc0[b] = v["w_fc0.%d" % b].data_ptr(), v["b_fc0.%d" % b].data_ptr()
            g.w_fc1[b], g.b_fc1[b] = v["w_fc1.%d" % b].data_ptr(), v["b_fc1.%d" % b].data_ptr()
        self.gc = g
        return g

    def unpack_grads(self) -> List[Optional[torch.Tensor]]:
        """Gradients in ``ResnetFC.ordered_params`` order (None if no chunk ran a backward)."""
        nb, H = self.n_blocks, self.d_hidden
        if self.gflat is None or not self.gviews:
            return [None] * (4 + 6 * nb)
        v = self.gviews
        out = [v["w_in"][:, :42], v["b_in"], v["w_out"][:self.d_out], v["b_out"][:self.d_out]]
        for b in range(nb):
            # lin_z.b.bias is added where lin_in.bias (b = 0) / fc_1.(b-1).bias is: the same column sums of dhz[b] (scenerf_hip.h)
            bz = v["b_in"] if b == 0 else v["b_fc1.%d" % (b - 1)]
            out += [v["w_fc0.%d" % b], v["b_fc0.%d" % b], v["w_fc1.%d" % b], v["b_fc1.%d" % b], v["w_z"][b * H:(b + 1) * H], bz]
        return out


class _GenericRun:
    """Buffers of one generic ResnetFC evaluation: what the rest of the chunk reads of an ``_MlpRun`` (+ the saved activations of a training pass)."""

    def __init__(self, M: int, pk: GenericPackedMLP, dev, keep_acts: bool = False):
        f32 = dict(dtype=torch.float32, device=dev)
        self.M = M
        self.Mpad = (M + _capi.TILE_ROWS - 1) // _capi.TILE_ROWS * _capi.TILE_ROWS
        self.sphere_idx = torch.empty((M, 2), dtype=torch.int32, device=dev)
        self.xenc = torch.empty((M, D_X), **f32)
        self.Z = torch.empty((self.Mpad, D_L), **f32)
        self.tile_mask = torch.empty((self.Mpad // _capi.TILE_ROWS,), dtype=torch.uint8, device=dev)
        self.tap_texel = torch.empty((M, 5, 4), dtype=torch.int32, device=dev)
        self.tap_weight = torch.empty((M, 5, 4), **f32)
        self.logits_pad = torch.empty((M, pk.d_out_pad), **f32)
        self.logits = None
        self.sign_bits = None
        self.h0pre = None
        self.acts = None
        if keep_acts:      # what scenerf_hip_resnetfc_backward reads: hz[b], n[b] per block, h_fin (scenerf_resnetfc_acts)
            nb = pk.n_blocks
            self.saved = [torch.empty((M, pk.d_hidden), **f32) for _ in range(2 * nb + 1)]
            self.h = [torch.empty((M, pk.d_hidden), **f32)]
            a = _capi.ResnetFCActs()
            for b in range(nb):
                a.hz[b], a.n[b] = self.saved[2 * b].data_ptr(), self.saved[2 * b + 1].data_ptr()
            a.h_fin = self.saved[2 * nb].data_ptr()
            self.acts = a
        else:
            self.h = [torch.empty((M, pk.d_hidden), **f32) for _ in range(3)]


class MlpHolder:
    def __init__(self, grad_sync=None, grad_sync_async=None):
        self.packed: Optional[PackedMLP] = None
        self.grad_sync = grad_sync   # callable(flat fp32 tensor) -> None, e.g. an all-reduce-mean (see dist.py)
        self.grad_sync_async = grad_sync_async   # callable(flat) -> finisher() or None: same collective, started early (dist.py)
        self.pending = None          # finisher of a collective in flight on the sink
        self.synced = False          # the sink was already reduced (early, on the side stream: RenderChunk.backward)
        self.single_chunk = False    # set by render_rays_batch when the whole call is one chunk
        self.defer_pack = False      # pack on the side stream (the radiance MLP: first used after the gaussian head's chain)
        self.split_pack = False      # ... in two calls, a forward's operands first (the gaussian head: its forward is the step's first GEMM)
        self.grad_mode = True        # torch.is_grad_enabled() where the session was opened (inside Function.forward it is always off)
        self.pack_cache: Optional[dict] = None   # caller-owned, valid while the parameters cannot change (one trainer forward): see PackMLP


def _pack_key(params, cfg, dev):
    cid = C.c_ulonglong(0)
    _capi.check(_capi.load().scenerf_hip_stream_capture_id(_stream(dev), C.byref(cid)), "stream_capture_id")
    return (cfg.precision_code, cid.value, tuple((id(t), t._version, t.data_ptr()) for t in params))


class PackMLP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, holder: MlpHolder, d_out: int, cfg: RenderConfig, *params):
        # training sessions (a parameter gradient will be asked for): the radiance MLP is packed on the side stream
        side = _side_stream(params[0].device) if (holder.defer_pack and any(ctx.needs_input_grad[3:])) else None
        if len(params) != len(MLP_PARAM_NAMES) or params[0].shape[0] != D_H:
            # any other ResnetFC shape: the per-layer fp32 GEMM path (scenerf_hip_resnetfc_forward / _forward_train / _backward)
            # (needs_input_grad reports the parameters' requires_grad flags whatever the grad mode: the session recorded the mode)
            # (trainable whenever the session runs with autograd on: the maps' gradients pass through this net too)
            holder.packed = GenericPackedMLP(params, d_out, cfg, trainable=bool(holder.grad_mode))
            ctx.holder = holder
            return torch.empty(1, device=params[0].device)
        cache, key = holder.pack_cache, None
        if cache is not None:
            # the trainer's per-image step opens 2 S sessions on the same parameters (S trained renders, S metric-only renders under
            # no_grad: scenerf.py:154-201): a metric-only session reads the operands the last session packed instead of packing them
            # again (3 + 3 launches, 44 MB written per session).  Only inside the caller's scope (training.TrainingMixin.forward: no
            # optimizer step in there), and still keyed by tensor identity, version counter, address and stream capture
            key = _pack_key(params, cfg, params[0].device)
            ent = cache.get(d_out)
            if not holder.grad_mode and ent is not None and ent[0] == key:
                pk = ent[1]
                pk.launch_pack()
                if pk._packed_ev is not None:
                    torch.cuda.current_stream(pk.device).wait_event(pk._packed_ev)
                holder.packed = pk
                ctx.holder = holder
                return torch.empty(1, device=params[0].device)
        holder.packed = PackedMLP(params, d_out, cfg, pack_stream=side, defer=CHAIN_FIRST, split=CHAIN_FIRST and holder.split_pack)
        if cache is not None:
            if side is None:
                holder.packed._packed_ev = torch.cuda.current_stream(params[0].device).record_event()
            cache[d_out] = (key, holder.packed)
        ctx.holder = holder
        return torch.empty(1, device=params[0].device)   # autograd token: its value is never read (no fill launch)

    @staticmethod
    def backward(ctx, _g):
        pk: PackedMLP = ctx.holder.packed
        if getattr(ctx, "done", False):
            raise RuntimeError("scenerf_amd: the MLP gradients of this render_rays_batch session were already handed to autograd "
                               "(a second backward over the same graph is not supported; call render_rays_batch again)")
        ctx.done = True
        with _on(pk.device):
            return PackMLP._backward(ctx, pk)

    @staticmethod
    def _backward(ctx, pk):
        # data-parallel hook: reduce the packed fp32 gradient sink in ONE collective (21.7 MB per MLP) before it is
        # carved into per-parameter views -- no flatten/unflatten copies, one large message per MLP over xGMI
        if ctx.holder.pending is not None:      # started in RenderChunk.backward, overlapped with the feature-gradient scatter
            ctx.holder.pending()
            ctx.holder.pending = None
        elif ctx.holder.grad_sync is not None and pk.gflat is not None and not ctx.holder.synced:
            ctx.holder.grad_sync(pk.gflat)
        ctx.holder.synced = False
        grads = pk.unpack_grads()
        grads = [g if ctx.needs_input_grad[3 + i] else None for i, g in enumerate(grads)]
        pk.gflat, pk.gc, pk.gviews = None, None, {}
        return (None, None, None) + tuple(grads)


# ------------------------------------------------------------------------------------------------ one MLP evaluation
class _MlpRun:
    """Buffers of one ResnetFC evaluation over M rows (kept for backward when grad is enabled)."""

    def __init__(self, M: int, d_out: int, prec: int, dev, lean: bool = False, x3_direct: bool = False, keep_xenc: bool = False):
        act = _act_dtype(prec)
        # lean = inference (no_grad) on the fused bf16 path (RenderConfig.uses_fused): lin_out runs inside the kernel, so no activation
        # is ever read again: neither they nor the sign bits are allocated or written (NULL in scenerf_mlp_acts)
        if lean and prec != 1:
            raise ValueError("lean activation buffers exist only on the fused bf16 path")
        self.M = M
        self.Mpad = (M + _capi.TILE_ROWS - 1) // _capi.TILE_ROWS * _capi.TILE_ROWS
        self.sphere_idx = torch.empty((M, 2), dtype=torch.int32, device=dev)
        # the fp32 encoding [M][48].  x3_direct (bf16 mode): encode_points writes the split-bf16 form straight into h0pre, the forward
        # neither reads this buffer nor launches the split -- it then exists only if asked for (keep_xenc: stage tests)
        if x3_direct and prec != 1:
            raise ValueError("the split encoding exists in bf16 mode only")
        self.x3_direct = x3_direct
        self.xenc = torch.empty((M, D_X), dtype=torch.float32, device=dev) if (not x3_direct or keep_xenc) else None
        self.Z = torch.empty((self.Mpad, D_L), dtype=act, device=dev)
        self.tile_mask = torch.empty((self.Mpad // _capi.TILE_ROWS,), dtype=torch.uint8, device=dev)
        self.tap_texel = torch.empty((M, 5, 4), dtype=torch.int32, device=dev)
        self.tap_weight = torch.empty((M, 5, 4), dtype=torch.float32, device=dev)
        self.H = [None if lean else torch.empty((M, D_H), dtype=act, device=dev) for i in range(4)]
        self.Nn = [None if lean else torch.empty((M, D_H), dtype=act, device=dev) for _ in range(3)]
        # fp32 lin_in output (fp32 mode) or the split-bf16 encoding [M][144] (bf16 mode)
        # (bf16: one row of slack -- lin_in's weight gradient reads the split encoding in 256-column tiles of its 144-column rows)
        self.h0pre = torch.empty((M, D_H) if prec == 0 else (M + 1, 3 * D_X // 2), dtype=torch.float32, device=dev)
        if prec == 1 and not lean and not x3_direct:
            self.h0pre[M:].zero_()    # (scenerf_hip.h: the slack row is read -- into scratch columns of the gradient sink -- and must be finite;
                                      #  scenerf_hip_encode_points zeroes it itself when it writes the split encoding: x3_direct)
        self.logits = torch.empty((M, d_out), dtype=torch.float32, device=dev)
        a = _capi.MlpActs()
        for i in range(4):
            a.H[i] = self.H[i].data_ptr() if self.H[i] is not None else None
        for i in range(3):
            a.Nn[i] = self.Nn[i].data_ptr() if self.Nn[i] is not None else None
        a.h0pre = self.h0pre.data_ptr()
        a.logits = self.logits.data_ptr()
        # sign bits of the seven saved activations (fused forward -> fused backward chain, scenerf_hip.h)
        self.sign_bits = torch.empty((7, self.Mpad, 64), dtype=torch.uint8, device=dev) if (prec and not lean) else None
        a.sign_bits = self.sign_bits.data_ptr() if self.sign_bits is not None else None
        a.x3_ready = 1 if x3_direct else 0
        self.c = a


def _mlp_eval(ccfg, cfg: RenderConfig, maps: MapHolder, pk: PackedMLP, dist, dist_ray_stride, ppr, unit_dir, viewdir,
              K, inv_K, T, M, keep_acts: bool = True, before_forward=None, before_wait=None) -> _MlpRun:
    lib = _capi.load()
    st = _stream(dist.device)
    if getattr(pk, "generic", False):
        keep_acts = keep_acts and pk.trainable
        run = _GenericRun(M, pk, dist.device, keep_acts=keep_acts)
        _capi.check(lib.scenerf_hip_encode_points(C.byref(ccfg), dist.data_ptr(), dist_ray_stride, ppr, unit_dir.data_ptr(),
                                                  viewdir.data_ptr(), K.data_ptr(), inv_K.data_ptr(), T.data_ptr(), M, None,
                                                  run.sphere_idx.data_ptr(), run.xenc.data_ptr(), None, st), "encode_points")
        _capi.check(lib.scenerf_hip_gather_features(C.byref(ccfg), C.byref(maps.map_ptr_array()), run.sphere_idx.data_ptr(), M,
                                                    run.Z.data_ptr(), run.tile_mask.data_ptr(), run.tap_texel.data_ptr(),
                                                    run.tap_weight.data_ptr(), st), "gather_features")
        if before_wait is not None:
            before_wait()
        if before_forward is not None:
            before_forward()
        if keep_acts:
            _capi.check(lib.scenerf_hip_resnetfc_forward_train(C.byref(ccfg), C.byref(pk.c), run.xenc.data_ptr(), run.Z.data_ptr(),
                                                               run.tile_mask.data_ptr(), M, C.byref(run.acts), run.h[0].data_ptr(),
                                                               run.logits_pad.data_ptr(), st), "resnetfc_forward_train")
        else:
            _capi.check(lib.scenerf_hip_resnetfc_forward(C.byref(ccfg), C.byref(pk.c), run.xenc.data_ptr(), run.Z.data_ptr(),
                                                         run.tile_mask.data_ptr(), M, run.h[0].data_ptr(), run.h[1].data_ptr(),
                                                         run.h[2].data_ptr(), run.logits_pad.data_ptr(), st), "resnetfc_forward")
        run.logits = run.logits_pad[:, :pk.d_out].contiguous()     # (the tail and the sampler read [M][d_out])
        run.h = None
        return run
    run = _MlpRun(M, pk.d_out, cfg.precision_code, dist.device, lean=(not keep_acts) and cfg.uses_fused(M),
                  x3_direct=cfg.precision_code == 1, keep_xenc=maps.debug_aux is not None)
    _capi.check(lib.scenerf_hip_encode_points(C.byref(ccfg), dist.data_ptr(), dist_ray_stride, ppr, unit_dir.data_ptr(),
                                              viewdir.data_ptr(), K.data_ptr(), inv_K.data_ptr(), T.data_ptr(), M, None,
                                              run.sphere_idx.data_ptr(), _capi.ptr(run.xenc),
                                              run.h0pre.data_ptr() if run.x3_direct else None, st), "encode_points")
    _capi.check(lib.scenerf_hip_gather_features(C.byref(ccfg), C.byref(maps.map_ptr_array()), run.sphere_idx.data_ptr(), M,
                                                run.Z.data_ptr(), run.tile_mask.data_ptr(), run.tap_texel.data_ptr(),
                                                run.tap_weight.data_ptr(), st), "gather_features")
    if before_wait is not None:
        before_wait()
    pk.wait_ready()   # (the operands may have been packed on the side stream: first needed here, behind the encode and the gather)
    if before_forward is not None:
        before_forward()
    _capi.check(lib.scenerf_hip_mlp_forward(C.byref(ccfg), C.byref(pk.c), run.Z.data_ptr(), _capi.ptr(run.xenc),
                                            run.tile_mask.data_ptr(), M, C.byref(run.c), st), "mlp_forward")
    return run


def _mlp_backward(ccfg, cfg: RenderConfig, maps: MapHolder, pk: PackedMLP, run: _MlpRun, d_logits, want_map_grads: bool,
                  sync_async=None, before_grads=None, maps_stream=None):
    """``sync_async`` (data parallel, scenerf_amd.dist.allreduce_mean_async): the parameter gradients are final once the weight-
    gradient GEMMs are queued, so their all-reduce is started there and the feature-gradient GEMM + scatter (0.5 ms) runs while
    the collective is in flight; returns the collective's finisher (or None).  ``before_grads``: called between the dgrad chain and the
    weight / feature-map gradients (scenerf_hip_mlp_backward's two-call form): where the caller orders that phase behind other work.
    ``maps_stream``: run the feature-map gradients THERE, beside the weight gradients on the current stream, and do not join: the map
    accumulators' next consumer waits (MapHolder._gmaps_ready -> grad_accumulators(): the next chunk's backward or PrepareMaps.backward) --
    the current stream then carries only what the PARAMETER gradients need, so an optimizer step ordered behind ``param_grads_ready``
    does not wait for the map gradients' tail (GraphedStep)."""
    lib = _capi.load()
    pk.wait_ready(backward=True)
    if getattr(pk, "generic", False):
        # any other ResnetFC shape: one fp32 GEMM per gradient, all on the current stream (scenerf_hip_resnetfc_backward)
        if before_grads is not None:
            before_grads()
        dev = d_logits.device
        H, nb = pk.d_hidden, pk.n_blocks
        dlog16 = torch.zeros((run.M, 16), dtype=torch.float32, device=dev)
        dlog16[:, :pk.d_out] = d_logits
        dhz = torch.empty((run.M, nb * H), dtype=torch.float32, device=dev)
        dh, dn = torch.empty((run.M, H), dtype=torch.float32, device=dev), torch.empty((run.M, H), dtype=torch.float32, device=dev)
        g = pk.grad_sink()
        gm = C.byref(maps.gmap_ptr_array()) if want_map_grads else None
        _capi.check(lib.scenerf_hip_resnetfc_backward(C.byref(ccfg), C.byref(pk.c), C.byref(pk.ct), C.byref(g), run.xenc.data_ptr(),
                                                      run.Z.data_ptr(), run.tile_mask.data_ptr(), run.tap_texel.data_ptr(),
                                                      run.tap_weight.data_ptr(), run.M, C.byref(run.acts), dlog16.data_ptr(), dhz.data_ptr(),
                                                      dh.data_ptr(), dn.data_ptr(), gm, _stream(dev)), "resnetfc_backward")
        return sync_async(pk.gflat) if sync_async is not None else None
    act = _act_dtype(cfg.precision_code)
    dev = d_logits.device
    dH = torch.empty((run.M, 4 * D_H), dtype=act, device=dev)
    dN = torch.empty((3, run.M, D_H), dtype=act, device=dev)
    g = pk.grad_sink()
    # (scatter on maps_stream: in that stream's order behind the previous chunk's -- the current stream does not wait for it)
    gm = C.byref(maps.gmap_ptr_array(wait=maps_stream is None)) if want_map_grads else None
    split = (sync_async is not None or maps_stream is not None) and gm is not None

    def call(cc):
        _capi.check(lib.scenerf_hip_mlp_backward(C.byref(cc), C.byref(pk.c), C.byref(g), run.Z.data_ptr(), _capi.ptr(run.xenc),
                                                 run.tile_mask.data_ptr(), run.tap_texel.data_ptr(), run.tap_weight.data_ptr(),
                                                 run.M, C.byref(run.c), d_logits.data_ptr(), dH.data_ptr(), dN.data_ptr(),
                                                 None if split else gm, _stream(dev)), "mlp_backward")

    chain_done = None
    if before_grads is None:
        call(ccfg)
    else:
        for flag in (_capi.FLAG_BWD_CHAIN_ONLY, _capi.FLAG_BWD_GRADS_ONLY):
            cc = type(ccfg).from_buffer_copy(ccfg)
            cc.flags |= flag
            if flag == _capi.FLAG_BWD_GRADS_ONLY:
                before_grads()
                if maps_stream is not None:      # what the feature-map gradients wait for: the chain (and the join), NOT the weight gradients
                    chain_done = torch.cuda.current_stream(dev).record_event()
            call(cc)
    finish = sync_async(pk.gflat) if sync_async is not None else None
    if split and maps_stream is not None:
        if chain_done is not None:
            maps_stream.wait_event(chain_done)
        else:
            maps_stream.wait_stream(torch.cuda.current_stream(dev))
        ccf = type(ccfg).from_buffer_copy(ccfg)
        ccf.flags |= _capi.FLAG_WGRAD_OVERLAP
        with torch.cuda.stream(maps_stream):
            _capi.check(lib.scenerf_hip_mlp_feature_grads(C.byref(ccf), C.byref(pk.c), run.tile_mask.data_ptr(), run.tap_texel.data_ptr(),
                                                          run.tap_weight.data_ptr(), run.M, dH.data_ptr(), gm, maps_stream.cuda_stream),
                        "mlp_feature_grads")
            maps._gmaps_ready = maps_stream.record_event()
        for t in (dH, run.tile_mask, run.tap_texel, run.tap_weight):
            t.record_stream(maps_stream)
    elif split:
        _capi.check(lib.scenerf_hip_mlp_feature_grads(C.byref(ccfg), C.byref(pk.c), run.tile_mask.data_ptr(), run.tap_texel.data_ptr(),
                                                      run.tap_weight.data_ptr(), run.M, dH.data_ptr(), gm, _stream(dev)),
                    "mlp_feature_grads")
    return finish


# ------------------------------------------------------------------------------------------------ one chunk of rays
class RenderChunk(torch.autograd.Function):
    """scenerf.py:598-700 for one chunk; differentiable w.r.t. the maps and both MLPs (through the tokens)."""

    @staticmethod
    def forward(ctx, cfg: RenderConfig, maps: MapHolder, mlp: MlpHolder, mlpg: MlpHolder, pixels, cam_K, inv_K, T_s2i,
                noise_u, noise_g, tok_maps, tok_mlp, tok_mlpg):
        with _on(pixels.device):
            return RenderChunk._forward(ctx, cfg, maps, mlp, mlpg, pixels, cam_K, inv_K, T_s2i, noise_u, noise_g)

    @staticmethod
    def _forward(ctx, cfg, maps, mlp, mlpg, pixels, cam_K, inv_K, T_s2i, noise_u, noise_g):
        ctx.set_materialize_grads(False)   # outputs nobody differentiates arrive as None, not as freshly filled zero tensors
        lib = _capi.load()
        dev = pixels.device
        st = _stream(dev)
        ccfg = cfg.to_c()
        R = pixels.shape[0]
        U, G, P, N = cfg.n_uni_used, cfg.n_gaussians, cfg.n_pts_per_gaussian, cfg.n_samples
        f32 = dict(dtype=torch.float32, device=dev)
        pixels, K, iK, T = _f32c(pixels), _f32c(cam_K), _f32c(inv_K), _f32c(T_s2i)
        # device_rng sessions (RenderSession.rng): neither noise tensor is drawn by torch -- ray_setup and gaussian_sample_sort make the
        # uniform / normal noise themselves from a {seed, call counter} pair in device memory (scenerf_hip.h)
        rng = maps.rng if (noise_u is None and noise_g is None and cfg.device_rng) else None
        if rng is None and noise_u is None and U > 0:
            raise RuntimeError("RenderChunk: noise_u is missing (only a device_rng session with an rng state may omit it)")
        noise_u = _f32c(noise_u).reshape(R, max(U, 0)) if (U > 0 and noise_u is not None) else None
        # constants the reference builds with torch.linspace (utils.py:79-81, scenerf.py:556-560)
        lin_u, anchors = _sampler_constants(dev, U, G, float(cfg.max_sample_depth))

        # device-drawn sampler noise without an rng state (a caller that injected only noise_u): requested NOW, in stream order -- between the
        # head's forward and the sampler it was a launch on the critical chain; on the side stream it was worse: in a replayed hipGraph
        # every edge between two queues costs 12-15 us of idle time (profiles/r04_g_step_trace.md), more than the 5 us the draw takes
        if noise_g is None and cfg.device_rng and rng is None:
            noise_g = draw_noise_g(cfg, R, dev)
        unit_dir = torch.empty((R, 3), **f32)
        viewdir = torch.empty((R, 3), **f32)
        dist_u = torch.empty((R, U), **f32) if U > 0 else None
        _capi.check(lib.scenerf_hip_ray_setup(C.byref(ccfg), pixels.data_ptr(), iK.data_ptr(), T.data_ptr(), _capi.ptr(lin_u),
                                              _capi.ptr(noise_u), _capi.ptr(rng), R, unit_dir.data_ptr(), viewdir.data_ptr(),
                                              _capi.ptr(dist_u), st), "ray_setup")
        # gaussian head on the G anchors per ray (scenerf.py:549-596)
        # (grad mode is off inside Function.forward: whether a backward can follow is what needs_input_grad says)
        keep = any(ctx.needs_input_grad)
        def _prefill4():
            if PREFILL_AT == 4 and getattr(maps, "_want_prefill", False):
                maps._want_prefill = False
                maps.prefill_grad_accumulators()
        fill5 = []
        def _launch_packs():   # deferred packs (CHAIN_FIRST): on the side stream, the head's first, behind the chain's first three launches
            fill6 = PREFILL_AT == 6 and getattr(maps, "_want_prefill", False) and getattr(maps, "_fill_after", None) is not None
            mlpg.packed.launch_pack(upto=1 if fill6 else 2)
            if fill6:
                # the fill in the side stream's own order, behind the half of the head's pack its forward waits for: it starts when the
                # head's encode and gather are over, runs beside the head's forward and is gone before the sampler; ordered behind the
                # point of the session's start where the accumulators were allocated (PrepareMaps.forward), not behind this step's chain
                maps._want_prefill = False
                maps.prefill_grad_accumulators(after=maps._fill_after)
                maps._fill_after = None
            mlp.packed.launch_pack()
            mlpg.packed.launch_pack()      # (the head's second half -- its backward's operands -- last)
            if PREFILL_AT == 5 and getattr(maps, "_want_prefill", False):     # fill beside the head's forward: fork point = behind its gather
                maps._want_prefill = False
                fill5.append(maps.alloc_grad_accumulators())
        run_g = _mlp_eval(ccfg, cfg, maps, mlpg.packed, anchors, 0, G, unit_dir, viewdir, K, iK, T, R * G, keep, before_forward=_prefill4,
                          before_wait=_launch_packs)
        if fill5 and fill5[0] is not None:
            maps.prefill_grad_accumulators(after=fill5[0])
        fill_after = None
        if PREFILL_AT in (1, 6) and getattr(maps, "_want_prefill", False):     # (6 without deferred packs: like 1)
            maps._want_prefill = False
            if CHAIN_FIRST:
                fill_after = maps.alloc_grad_accumulators()    # (the fill itself: behind the sampler's launch, below)
            else:
                maps.prefill_grad_accumulators()
        # the gaussian sampler's normal noise, if the caller did not inject it: drawn HERE, with the gaussian head's chain already queued
        # -- the reference's host-side draw (utils.py:208-211) takes ~0.25 ms of host time per 1,200 rays, which at the top of the
        # chunk left the GPU without work (3.29 -> 3.04 ms per KITTI step, tools/ab_host.py devrng); same generator, same call order
        if rng is not None:
            noise_g = torch.empty((R, G * P), **f32)     # written by the sampler (its backward reads it)
        elif noise_g is None:
            noise_g = draw_noise_g(cfg, R, dev)
        noise_g = _f32c(noise_g).reshape(R, G * P)
        gmeans = torch.empty((R, G), **f32)
        gstds = torch.empty((R, G), **f32)
        dist_s = torch.empty((R, N), **f32)
        z_s = torch.empty((R, N), **f32)
        perm = torch.empty((R, N), dtype=torch.int32, device=dev)
        _capi.check(lib.scenerf_hip_gaussian_sample_sort(C.byref(ccfg), run_g.logits.data_ptr(), anchors.data_ptr(),
                                                         _capi.ptr(dist_u), noise_g.data_ptr(), _capi.ptr(rng), unit_dir.data_ptr(), R,
                                                         gmeans.data_ptr(), gstds.data_ptr(), dist_s.data_ptr(), z_s.data_ptr(),
                                                         perm.data_ptr(), st), "gaussian_sample_sort")
        if fill_after is not None:
            maps.prefill_grad_accumulators(after=fill_after)
        # radiance MLP on the sorted samples (scenerf.py:661-665)
        # (its forward joins the side stream for the packed operands: the accumulator fill queued there is waited for by the same node)
        run_m = _mlp_eval(ccfg, cfg, maps, mlp.packed, dist_s, N, N, unit_dir, viewdir, K, iK, T, R * N, keep,
                          before_forward=maps.join_prefill if (CHAIN_FIRST and keep) else None)
        if PREFILL_AT == 2 and getattr(maps, "_want_prefill", False):
            maps._want_prefill = False
            maps.prefill_grad_accumulators()
        # a no_grad session that asked for a subset of the outputs (ImageRenderer(keys=...): RenderSession.want_keys): the (R, N) outputs
        # nobody reads are not written, and without loss_kl / som_vars the per-ray tail runs its compositing-only instantiation
        want = getattr(maps, "want_keys", None) if (not keep and maps.debug_aux is None) else None
        need = (lambda k: True) if want is None else (lambda k: k in want)
        som = need("loss_kl") or need("som_vars")
        nothing = torch.empty((0,), **f32)
        dens = torch.empty((R, N), **f32) if need("densities") else nothing
        alphas = torch.empty((R, N), **f32) if (need("alphas") or som) else nothing
        weights = torch.empty((R, N), **f32) if need("weights") else nothing
        depth = torch.empty((R,), **f32)
        color = torch.empty((R, 3), **f32)
        closest = torch.empty((R,), **f32)
        w_at = torch.empty((R,), **f32)
        closest_idx = torch.empty((R,), dtype=torch.int32, device=dev)
        loss_kl = torch.empty((R,), **f32) if som else nothing
        som_means = torch.empty((R, G), **f32) if som else nothing
        som_vars = torch.empty((R, G), **f32) if som else nothing
        kl_saved = torch.empty((R, G, 3), **f32) if som else nothing
        bmu = torch.empty((R, N), dtype=torch.uint8, device=dev) if maps.debug_aux is not None else None
        opt = lambda t: t.data_ptr() if t.numel() else None     # noqa: E731
        # compositing + RaySOM: one launch, the alphas stay in the wave's registers for the SOM update (scenerf_hip.h: ray_tail)
        _capi.check(lib.scenerf_hip_ray_tail_forward(C.byref(ccfg), run_m.logits.data_ptr(), dist_s.data_ptr(), z_s.data_ptr(),
                                                     gmeans.data_ptr(), gstds.data_ptr(), R, opt(dens), opt(alphas),
                                                     opt(weights), depth.data_ptr(), color.data_ptr(), closest.data_ptr(),
                                                     w_at.data_ptr(), closest_idx.data_ptr(), opt(loss_kl), opt(som_means),
                                                     opt(som_vars), opt(kl_saved), _capi.ptr(bmu), st), "ray_tail_forward")
        # keep what backward needs (plain attributes: these are internal buffers, not graph tensors)
        ctx.cfg, ctx.ccfg, ctx.maps, ctx.mlp, ctx.mlpg = cfg, ccfg, maps, mlp, mlpg
        ctx.keep = dict(R=R, anchors=anchors, noise_g=noise_g, unit_dir=unit_dir, gmeans=gmeans, gstds=gstds, perm=perm,
                        dist_s=dist_s, z_s=z_s, kl_saved=kl_saved, run_g=run_g, run_m=run_m, closest_idx=closest_idx)
        ctx.mark_non_differentiable(som_vars, som_means)    # (made from detached inputs in the reference too: ray_som_kl.py:17-19)
        if maps.debug_aux is not None:
            # opt-in debug hook (RenderSession(debug_aux=True)): stage intermediates of the LAST chunk of this session, for the
            # per-stage parity tests.  Off by default: it would pin logits / xenc / indices of a chunk (GBs at N = 512) for as long
            # as the session lives, and a process-wide slot would be shared between models and threads.
            maps.debug_aux.clear()
            maps.debug_aux.update(perm=perm, sphere_idx=run_m.sphere_idx, closest_idx=closest_idx, tile_mask=run_m.tile_mask,
                                  offsets=run_g.logits, logits=run_m.logits, dist_sorted=dist_s, xenc=run_m.xenc,
                                  sphere_idx_g=run_g.sphere_idx, unit_dir=unit_dir, viewdir=viewdir, dist_u=dist_u,
                                  tile_mask_g=run_g.tile_mask, bmu=bmu, kl_mask=kl_saved[:, :, 2], som_means=som_means)
        # order = OUTPUT_KEYS + som_means
        return depth, color, gmeans, gstds, w_at, closest, loss_kl, alphas, som_vars, dens, weights, z_s, som_means

    @staticmethod
    def backward(ctx, *grads):
        if ctx.keep is None:
            # the saved activations (GBs per chunk) are released by the first backward and the gradient sinks are handed to autograd
            # once per session: a second pass over the same graph cannot be served
            raise RuntimeError("scenerf_amd: this render_rays_batch graph was already back-propagated (retain_graph=True / a second "
                               "backward or autograd.grad over the same outputs is not supported: the saved activations are freed "
                               "by the first pass).  Call render_rays_batch again for another backward.")
        with _on(ctx.keep["dist_s"].device):
            return RenderChunk._backward(ctx, *grads)

    @staticmethod
    def _backward(ctx, g_depth, g_color, g_gmeans, g_gstds, g_wat, g_closest, g_kl, g_alphas, _g_somv, g_dens, g_weights,
                  g_zvol, _g_somm):
        lib = _capi.load()
        k = ctx.keep
        st = _stream(k["dist_s"].device)
        cfg, ccfg = ctx.cfg, ctx.ccfg
        R, N, G = k["R"], cfg.n_samples, cfg.n_gaussians
        dev = k["dist_s"].device
        f32 = dict(dtype=torch.float32, device=dev)

        def c(t):
            return None if t is None else _f32c(t)

        g_depth = c(g_depth) if g_depth is not None else torch.zeros((R,), **f32)
        g_color = c(g_color) if g_color is not None else torch.zeros((R, 3), **f32)
        g_gmeans, g_gstds, g_kl, g_alphas, g_dens, g_weights, g_zvol, g_wat, g_closest = map(c, (g_gmeans, g_gstds, g_kl, g_alphas, g_dens,
                                                                                                   g_weights, g_zvol, g_wat, g_closest))
        run_m, run_g = k["run_m"], k["run_g"]
        d_logits = torch.empty((R * N, 4), **f32)
        d_off = torch.empty((R, G, 2), **f32)
        # compositing backward + sampler / KL backward: one launch, d_dist / d_z never reach HBM (scenerf_hip.h: ray_tail)
        _capi.check(lib.scenerf_hip_ray_tail_backward(C.byref(ccfg), run_m.logits.data_ptr(), k["dist_s"].data_ptr(), k["z_s"].data_ptr(), R,
                                                      g_depth.data_ptr(), g_color.data_ptr(), _capi.ptr(g_weights), _capi.ptr(g_alphas),
                                                      _capi.ptr(g_dens), _capi.ptr(g_zvol), run_g.logits.data_ptr(),
                                                      k["anchors"].data_ptr(), k["noise_g"].data_ptr(), k["unit_dir"].data_ptr(),
                                                      k["gmeans"].data_ptr(), k["gstds"].data_ptr(), k["perm"].data_ptr(),
                                                      k["kl_saved"].data_ptr(), _capi.ptr(g_kl), _capi.ptr(g_gmeans), _capi.ptr(g_gstds),
                                                      d_logits.data_ptr(), d_off.data_ptr(), None, None, _capi.ptr(g_wat), _capi.ptr(g_closest),
                                                      k["closest_idx"].data_ptr(), st), "ray_tail_backward")
        want_maps = bool(ctx.needs_input_grad[10])
        if PREFILL_AT == 3 and want_maps and getattr(ctx.maps, "_want_prefill", False):
            ctx.maps._want_prefill = False
            ctx.maps.prefill_grad_accumulators()      # (side stream, behind the per-ray tail's backward: beside lin_out's reduction and the chain)
        # The gaussian head's backward (R*G rows: small grids) is independent of the radiance MLP's backward: run it
        # on a side stream so its workgroups fill the gaps of the big GEMMs.  Both scatter into the same map-gradient
        # accumulators with atomics; the main stream waits for the side stream before anything reads them.
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev)
        do_head = bool(ctx.needs_input_grad[12] or want_maps)
        head_done = []
        # (not with gradient collectives in the session: the head's all-reduce runs on the side stream in front of the scatter, and a
        #  captured step then replays the head's backward + all-reduce in FRONT of the chain -- one RCCL rank, replayed: 3.29 ms against
        #  2.62 with the scatter on the main stream, 2.59 issued eagerly; profiles/r06_u_*)
        #  -- issued eagerly the side-stream scatter stays: 2.59 against 2.71 ms)
        collectives = (ctx.mlpg.grad_sync is not None or ctx.mlp.grad_sync_async is not None or ctx.mlp.grad_sync is not None) \
            and torch.cuda.is_current_stream_capturing()
        side_maps = bool(MAP_GRADS_ON_SIDE and want_maps and GRADS_BEHIND_HEAD and do_head and ctx.needs_input_grad[11] and not collectives
                         and (ctx.mlpg.single_chunk or MAP_GRADS_ON_SIDE_MULTI))

        def main_backward():
            if ctx.needs_input_grad[11] or want_maps:
                early = ctx.mlp.grad_sync_async if (ctx.mlpg.single_chunk and ctx.needs_input_grad[11]) else None
                # GRADS_BEHIND_HEAD: the radiance MLP's weight / feature-map gradient phase waits for the gaussian head's backward (its
                # kernels, not its all-reduce) -- see SCENERF_FLAG_BWD_CHAIN_ONLY in csrc/mlp.hip for what happens when the two meet
                join = (lambda: main.wait_event(head_done[0])) if (GRADS_BEHIND_HEAD and head_done) else None
                # MAP_GRADS_ON_SIDE (single-chunk training sessions): the weight gradients stay on THIS stream, the feature-map
                # gradients go to the side stream (the head's backward there is over: joined above) and are not waited for here
                # (round 6, later: any number of chunks -- the S source frames of an image are chunks of one session in the trainer's step:
                #  a chunk's scatter follows the previous chunk's on the side stream, the chain of the next chunk runs beside its tail)
                on_side = side_maps and join is not None
                ccm = ccfg
                if MAIN_WGRAD_OVERLAP and not on_side:
                    ccm = type(ccfg).from_buffer_copy(ccfg)
                    ccm.flags |= _capi.FLAG_WGRAD_OVERLAP
                ctx.mlp.pending = _mlp_backward(ccm, cfg, ctx.maps, ctx.mlp.packed, run_m, d_logits, want_maps, sync_async=early,
                                                before_grads=join, maps_stream=side if on_side else None)
                # both MLPs' parameter gradients are complete at this point of this stream (the head's: joined above)
                if join is not None and getattr(ctx.maps, "events", None) is not None:
                    ctx.maps.events["param_grads_ready"] = main.record_event()

        def head_backward():
            side.wait_stream(main)
            with torch.cuda.stream(side):
                _mlp_backward(ccfg, cfg, ctx.maps, ctx.mlpg.packed, run_g, d_off.view(R * G, 2), want_maps)
                head_done.append(side.record_event())
                # data parallel, one chunk per step (training): the head's parameter gradients are final here, ~2 ms before the
                # radiance MLP's -- reduce them now, on the side stream, under the main backward (half of the step's all-reduce
                # volume leaves the critical path).  Every rank takes this branch in the same order: head first, main MLP later.
                if ctx.mlpg.single_chunk and ctx.mlpg.grad_sync is not None and ctx.needs_input_grad[12]:
                    ctx.mlpg.grad_sync(ctx.mlpg.packed.gflat)
                    ctx.mlpg.synced = True

        if do_head:
            if want_maps:
                # allocate + zero on the main stream before the fork; (side_maps: every scatter of this backward is launched on the side
                # stream, behind the previous chunk's -- the main stream has nothing to wait for)
                ctx.maps.grad_accumulators(wait=not side_maps)
            ctx.mlpg.packed.grad_sink()
            # (the head's backward is launched FIRST here although it is not the chain: launched second it shares the replayed graph's
            #  second queue with the radiance MLP's weight gradients, and that queue runs the deeper fork's kernels first -- the head's
            #  backward then waits ~0.6 ms behind kernels that are not ready (tools/graph_queue_probe.py: order_h_then_w / order_w_then_h;
            #  measured in the step: 2.93 against 2.67 ms).  Launched first it keeps the chain's queue and the chain moves on: 13 us
            #  at this fork instead of 5.)
            head_backward()
            main_backward()
            if ctx.mlpg.synced or not (GRADS_BEHIND_HEAD and head_done and (ctx.needs_input_grad[11] or want_maps)):
                main.wait_stream(side)       # (the head's all-reduce, or no join inside main_backward)
            for t in (d_off, run_g.Z, run_g.xenc, run_g.h0pre, run_g.logits):
                if t is not None:
                    t.record_stream(side)
        else:
            main_backward()
        ctx.keep = None
        z1 = _zero_token(dev)
        return (None, None, None, None, None, None, None, None, None, None,
                z1 if ctx.needs_input_grad[10] else None, z1 if ctx.needs_input_grad[11] else None,
                z1 if ctx.needs_input_grad[12] else None)


# ------------------------------------------------------------------------------------------------ public API
class RenderSession:
    """Per-call state of ``render_rays_batch``: converted maps + packed MLPs, shared by all chunks."""

    def __init__(self, cfg: RenderConfig, x_rgb: Dict[str, torch.Tensor], mlp_params: Sequence[torch.Tensor],
                 mlpg_params: Sequence[torch.Tensor], grad_sync=None, grad_sync_async=None, debug_aux: bool = False,
                 rng: Optional[torch.Tensor] = None, events: Optional[dict] = None, convert_cache: Optional[dict] = None,
                 pack_cache: Optional[dict] = None):
        """``events``: a caller-owned dict; a training session's backward leaves ``events["param_grads_ready"]`` there -- an event of the
        backward's stream behind which BOTH MLPs' parameter gradients are complete (the feature-map gradients may still be running on
        the side stream: whoever reads them goes through PrepareMaps.backward, which waits)."""
        hwc, chw = self.classify_maps(x_rgb)
        if hwc or cfg.hwc_scales:   # per-call layout state (scenerf_cfg.map_chw): a copy, the model's config is not touched
            cfg = dataclasses.replace(cfg, hwc_scales=hwc, direct_scales=tuple(i for i in cfg.direct_scales if i not in hwc))
        _require_cuda(chw[0], "x_rgb map 0")
        self.device = chw[0].device
        self._convert_cache = convert_cache
        self._pack_cache = pack_cache        # see PackMLP.forward: only a caller inside which the parameters cannot change passes one
        with _on(self.device):
            self._open(cfg, chw, mlp_params, mlpg_params, grad_sync, grad_sync_async, debug_aux)
        if rng is not None:
            if not (rng.is_cuda and rng.dtype == torch.int64 and rng.numel() == 3 and rng.device == self.device):
                raise RuntimeError("rng must be a CUDA int64 tensor {seed, calls, scratch} on the maps' device")
            self.maps.rng = rng if cfg.device_rng else None
        self.maps.events = events

    @staticmethod
    def classify_maps(x_rgb):
        """(indices of the levels that are read in place as (H,W,C), the five tensors as handed to PrepareMaps).  Channels-last maps
        are read in place: either wrapped (HWC: an (H,W,C) tensor) or a (C,H,W) tensor whose MEMORY is (H,W,C) -- a slice of a
        torch.channels_last (B,C,H,W) batch has exactly these strides -- which enters as its (H,W,C) view: autograd carries the
        gradient back through the permute, no copy in either direction."""
        vals = [x_rgb["1_%d" % s] for s in (1, 2, 4, 8, 16)]
        hwc, chw = [], []
        for i, v in enumerate(vals):
            if isinstance(v, HWC):
                hwc.append(i); chw.append(v.t)
            elif v.dim() == 3 and v.shape[0] > 1 and v.dtype == torch.float32 and v.stride() == (1, v.shape[2] * v.shape[0], v.shape[0]):
                hwc.append(i); chw.append(v.permute(1, 2, 0))
            else:
                chw.append(v)
        return tuple(hwc), chw

    def _open(self, cfg, chw, mlp_params, mlpg_params, grad_sync, grad_sync_async, debug_aux):
        lib = _capi.load()
        cfg.validate()
        self.cfg = cfg
        # per-device one-time setup (kernel attributes, descriptor tables): explicit, so that a later hipGraph capture of a chunk
        # never meets a first-use allocation
        _capi.check(lib.scenerf_hip_prepare(C.byref(cfg.to_c()), _stream(self.device)), "prepare")
        # autograd runs ready nodes newest-first: the MLP tokens are created BEFORE the map token so that in backward the map
        # transposes (PrepareMaps.backward) are queued before PackMLP.backward waits for a gradient all-reduce in flight
        self.mlp, self.mlpg = MlpHolder(grad_sync, grad_sync_async), MlpHolder(grad_sync)
        # training sessions pack both MLPs on the side stream: the gaussian head's operands (needed first) are packed while the main
        # stream sets up the rays and gathers the head's features (~55 us before its first GEMM), the radiance MLP's behind them
        # (first read ~0.3 ms into the step).  The side stream runs them in this order: head, then radiance MLP.
        self.mlp.grad_mode = self.mlpg.grad_mode = torch.is_grad_enabled()
        self.mlp.pack_cache = self.mlpg.pack_cache = self._pack_cache
        self.mlp.defer_pack = True
        self.mlpg.defer_pack = DEFER_HEAD_PACK
        self.mlpg.split_pack = SPLIT_HEAD_PACK
        self.tok_mlpg = PackMLP.apply(self.mlpg, 2, cfg, *mlpg_params)
        self.tok_mlp = PackMLP.apply(self.mlp, 4, cfg, *mlp_params)
        self.maps = MapHolder(cfg)
        self.maps.convert_cache = self._convert_cache
        self.maps.grad_mode = torch.is_grad_enabled()
        if debug_aux:
            self.maps.debug_aux = {}
        self.tok_maps = PrepareMaps.apply(self.maps, *chw)

    @property
    def last_aux(self) -> Dict[str, torch.Tensor]:
        """Stage intermediates of the last rendered chunk (only with ``debug_aux=True``)."""
        if self.maps.debug_aux is None:
            raise RuntimeError("open the RenderSession with debug_aux=True to keep stage intermediates")
        return self.maps.debug_aux

    def draw_noise(self, R: int, device):
        """The reference's in-path RNG calls, same generators and order (SURVEY §5 RNG row): (uniform noise, gaussian noise)."""
        return self._draw_noise_u(R, device), draw_noise_g(self.cfg, R, device)

    def _draw_noise_u(self, R: int, device):
        """utils.py:84 (torch.rand_like on the device) for the uniform samples the reference draws: n_pts_uni of them, or the variant's
        substitute when that is 0 (scenerf_bf.py:623-626) -- drawn even where they are not rendered (gaussian-only branch), so that the
        device generator advances exactly as in the reference."""
        cfg = self.cfg
        drawn = cfg.n_uni_drawn
        nu = torch.rand((R, drawn, 1), dtype=torch.float32, device=device) if drawn > 0 else None
        if cfg.n_uni_used == 0 or nu is None:
            return torch.empty((R, 0, 1), device=device)
        return nu

    def render_chunk(self, pixels, cam_K, inv_K, T_s2i, noise_u=None, noise_g=None) -> Dict[str, torch.Tensor]:
        _require_cuda(pixels, "sampled_pixels")
        if noise_u is None and not (self.maps.rng is not None and noise_g is None):
            # (the gaussian noise, if not injected, is drawn inside the chunk: RenderChunk._forward; a device_rng session with an rng
            #  state draws neither: the kernels make their noise)
            noise_u = self._draw_noise_u(pixels.shape[0], pixels.device)
        outs = RenderChunk.apply(self.cfg, self.maps, self.mlp, self.mlpg, pixels, cam_K, inv_K, T_s2i, noise_u, noise_g,
                                 self.tok_maps, self.tok_mlp, self.tok_mlpg)
        ret = dict(zip(OUTPUT_KEYS + ["som_means"], outs))
        return ret
