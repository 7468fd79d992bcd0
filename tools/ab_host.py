"""A/B of host-side scheduling choices on one GPU box (box-to-box spread is +-4 %, more than what is being measured): runs bench.py in
this process with parts of the renderer's stream plumbing switched off.
usage: ab_host.py <mode> [bench.py arguments]     mode: comma list of  base | nopre (map-gradient accumulators zeroed in the backward)
                                                                         | nodefer (radiance-MLP pack on the main stream)
                                                                         | devrng (sampling noise drawn on the device)
                                                                         | overlap (scenerf_cfg flag WGRAD_OVERLAP: weight gradients on the library's side stream)"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode = sys.argv[1].split(",")
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
import scenerf_amd.renderer as r
if "nopre" in mode:
    r.MapHolder.prefill_grad_accumulators = lambda self: None
if "nodefer" in mode:
    _init = r.PackedMLP.__init__
    def init(self, params, d_out, cfg, pack_stream=None):
        _init(self, params, d_out, cfg, pack_stream=None)
    r.PackedMLP.__init__ = init
if "devrng" in mode:       # the samplers' normal noise drawn on the device instead of on the host like the reference (RenderConfig.device_rng)
    from scenerf_amd import config
    config.RenderConfig.__init__.__kwdefaults__ and None
    _orig_init = config.RenderConfig.__init__
    def _init(self, *a, **k):
        k["device_rng"] = True
        _orig_init(self, *a, **k)
    config.RenderConfig.__init__ = _init
if "fillloop" in mode:     # the map-gradient accumulators zeroed one fill per tensor instead of one multi-tensor launch
    import torch
    def _loop_zero(ts):
        for t in ts:
            t.zero_()
    torch._foreach_zero_ = _loop_zero
if "overlap" in mode:
    from scenerf_amd import _capi, config
    _to_c = config.RenderConfig.to_c
    def to_c(self):
        c = _to_c(self)
        c.flags |= _capi.FLAG_WGRAD_OVERLAP
        return c
    config.RenderConfig.to_c = to_c
runpy.run_path(sys.argv[0], run_name="__main__")
