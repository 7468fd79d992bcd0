"""The trainer's per-image step (bench.kitti_training_step_leg: eager, then one hipGraph replay per step) on its own, for a kernel trace:
rocprofv3 --kernel-trace --output-format csv -d <dir> -- python tools/trainer_step_probe.py ; python tools/step_trace.py <dir>"""
import json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.argv = ["bench.py"] + sys.argv[1:]
import bench                                   # noqa: E402
from scenerf_amd import _capi                  # noqa: E402

args = bench.parse()
_capi.load()
o = bench.kitti_training_step_leg(args, torch.device("cuda", 0), steps=int(os.environ.get("PROBE_STEPS", "8")))
print(json.dumps({k: o[k] for k in ("eager", "graphed")}))
