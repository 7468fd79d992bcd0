"""Micro-probe of the NT/TN GEMM kernels through the C ABI test entry points (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scenerf_amd import _capi

lib = _capi.load()
dev = "cuda"

def time_nt(M, N, K, tile, prec=1, iters=20):
    dt = torch.bfloat16 if prec else torch.float32
    A = torch.randn(M, K, device=dev).to(dt)
    W = (torch.randn(N, K, device=dev) * 0.05).to(dt)
    C = torch.empty(M, N, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        lib.scenerf_hip_test_gemm_nt(prec, A.data_ptr(), W.data_ptr(), None, M, N, K, 1, tile, C.data_ptr(), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.scenerf_hip_test_gemm_nt(prec, A.data_ptr(), W.data_ptr(), None, M, N, K, 1, tile, C.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    return us, 2.0 * M * N * K / us / 1e6

def time_tn(M, N, K, prec=1, iters=20):
    dt = torch.bfloat16 if prec else torch.float32
    D = (torch.randn(M, N, device=dev) * 0.05).to(dt)
    A = torch.randn(M, K, device=dev).to(dt)
    C = torch.zeros(N, K, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        lib.scenerf_hip_test_gemm_tn(prec, D.data_ptr(), A.data_ptr(), M, N, K, 1, C.data_ptr(), None, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.scenerf_hip_test_gemm_tn(prec, D.data_ptr(), A.data_ptr(), M, N, K, 1, C.data_ptr(), None, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    return us, 2.0 * M * N * K / us / 1e6

if __name__ == "__main__":
    M = 153600
    for (N, K, tile) in [(512, 512, 3), (512, 1024, 3), (512, 1536, 3), (512, 512, 4), (512, 1024, 4), (512, 1536, 4)]:
        us, tf = time_nt(M, N, K, tile)
        print("NT bf16 M=%d N=%d K=%d tile=%d: %8.1f us  %7.1f TF/s" % (M, N, K, tile, us, tf))
    for (N, K) in [(512, 512), (1536, 80)]:
        us, tf = time_tn(M, N, K)
        print("TN bf16 M=%d N=%d K=%d: %8.1f us  %7.1f TF/s" % (M, N, K, us, tf))
    us, tf = time_nt(M, 512, 512, 2, prec=0)
    print("NT fp32 wide: %8.1f us %7.1f TF/s" % (us, tf))
