"""One NT (wide) + one TN launch config, few iterations: target for rocprofv3 --pmc runs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_probe import time_nt, time_tn
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 2
print(time_nt(153600, 512, 512, tile, iters=5))
print(time_tn(153600, 512, 512, iters=5))
