// Micro-benchmark: do two waves on one SIMD overlap one wave's scalar/vector bookkeeping with the other wave's MFMAs?
// Each wave loops over [NM MFMAs on independent accumulators] + [NV dependent-free VALU adds + NS SALU adds]; 1 or 2 waves per SIMD.
// Perfect overlap: time per iteration per SIMD = max(waves * NM * 32, issue time); none: waves * (NM * 32 + (NV + NS) * ~4..8).
//   hipcc --offload-arch=gfx950 -O3 mfma_overlap.hip -o mfma_overlap && ./mfma_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NM, int NV, int NS>
__global__ __launch_bounds__(512) void k(int iters, float* sink, int seed) {
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 3); b[e] = (__bf16)1.0f; }
    int v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    int s0 = seed, s1 = seed + 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 7], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NV; ++i) { v[i & 7] = v[i & 7] * 3 + it; asm volatile("" : "+v"(v[i & 7])); }
#pragma unroll
        for (int i = 0; i < NS; ++i) { s0 = s0 * 5 + s1; asm volatile("" : "+s"(s0)); s1 ^= s0; asm volatile("" : "+s"(s1)); }
        __builtin_amdgcn_sched_barrier(0);
    }
    float t = (float)(s0 + s1);
#pragma unroll
    for (int i = 0; i < 8; ++i) { t += acc[i][0] + acc[i][7]; t += (float)v[i]; }
    if (t == 12345.678f) sink[0] = t;
}

template <int NM, int NV, int NS> void run(float* sink, int threads) {
    const int iters = 4000, wgs = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NM, NV, NS><<<wgs, threads>>>(iters, sink, 1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<NM, NV, NS><<<wgs, threads>>>(iters, sink, 1);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns_it = ms * 1e6 / iters;
    printf("NM=%d NV=%2d NS=%2d, %d waves/SIMD: %.1f ns per iteration = %.0f cycles @2.4GHz ; MFMA-bound would be %d cycles\n", NM, NV, NS,
           threads / 256, ns_it, ns_it * 2.4, (threads / 256) * NM * 32);
}

int main() {
    float* sink; (void)hipMalloc(&sink, 4);
    run<8, 0, 0>(sink, 256);  run<8, 0, 0>(sink, 512);
    run<8, 24, 16>(sink, 256); run<8, 24, 16>(sink, 512);
    run<8, 48, 32>(sink, 256); run<8, 48, 32>(sink, 512);
    run<4, 24, 16>(sink, 256); run<4, 24, 16>(sink, 512);
    return 0;
}
