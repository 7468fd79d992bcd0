"""One NT launch config, few iterations: target for rocprofv3 --pmc runs.  usage: gemm_probe_one.py TILE K"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_probe import time_nt, time_tn
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
print(time_nt(153600, 512, K, tile, iters=5))
