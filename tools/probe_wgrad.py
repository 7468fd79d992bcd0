"""Time the weight-gradient GEMM (new transposing-read kernel vs gemm_tn_kernel) at the bench shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = sys.argv[:1]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_probe.py")).read().split("if __name__")[0])
for (M, N, K) in ((153600, 512, 512), (153600, 256, 256), (40000, 512, 512)):
    us, tf = time_tn(M, N, K)
    print("TN M=%d N=%d K=%d: %.1f us %.0f TF/s  (SRF_NO_WGRAD_TR=%s, WGS=%s)" % (M, N, K, us, tf, os.environ.get("SRF_NO_WGRAD_TR"), os.environ.get("SRF_WGRAD_TR_WGS")))
