// Issue rate of v_mfma_f32_32x32x16_bf16: 8 independent accumulators per wave (the weight-gradient kernel's wave tile), accumulators in
// architectural VGPRs or in AGPRs, 1 / 2 / 3 waves per SIMD.  Prints shader cycles per MFMA per SIMD.
// hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int MODE>
__global__ __launch_bounds__(768) void k(int iters, unsigned long long* out, float* sink) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 7); b[e] = (__bf16)1.0f; }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0 && threadIdx.x < 256) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 12345.f) sink[0] = s;
}
template <int MODE> void run(const char* name, int waves) {
    unsigned long long* out; float* sink;
    (void)hipMalloc((void**)&out, 8 * 4 * 256); (void)hipMalloc((void**)&sink, 4);
    const int iters = 2000;
    k<MODE><<<256, waves * 64>>>(iters, out, sink);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k<MODE><<<256, waves * 64>>>(iters, out, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[1024]; (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0; for (int i = 0; i < 1024; ++i) cyc += h[i]; cyc /= 1024;
    const double per_simd = (double)iters * 8 * (waves / 4);
    printf("%-10s %2d waves/CU: %.1f cycles per MFMA per SIMD; %.3f ms -> counter at %.2f GHz; %.0f TFLOP/s on 256 CUs\n", name, waves, cyc / per_simd, ms,
           cyc / ms / 1e6, 256.0 * waves * iters * 8 * 32768.0 / ms / 1e9);
    (void)hipFree(out); (void)hipFree(sink);
}
int main() {
    run<0>("acc in VGPR", 4); run<1>("acc in AGPR", 4);
    run<0>("acc in VGPR", 8); run<1>("acc in AGPR", 8);
    run<0>("acc in VGPR", 12); run<1>("acc in AGPR", 12);
    return 0;
}
