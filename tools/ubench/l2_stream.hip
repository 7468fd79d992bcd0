// Micro-benchmark: how fast can every CU stream the SAME L2-resident weight panel into LDS with global_load_lds?
// (feasibility probe for a fused row-block MLP kernel whose only global traffic in the main loop is W from L2)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__device__ static inline void glds16(const void* g, unsigned lds_wave_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_wave_base) : "memory");
}

// each workgroup (NT threads) streams `bytes` of W `iters` times through a ring of 32 KiB LDS stages
template <int NT>
__global__ __launch_bounds__(NT) void stream_kernel(const char* W, size_t bytes, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
    constexpr int NW = NT / 64;
    constexpr int STAGE = 32768, NSTAGE = 3;
    const int pieces_per_stage = STAGE / 1024;          // 1 KiB per wave-instruction
    const int ppw = pieces_per_stage / NW;              // pieces per wave per stage
    const size_t nst = bytes / STAGE;
    float acc = 0.f;
    size_t total = nst * iters;
    // prologue: 2 stages in flight
    for (int s = 0; s < 2; ++s) {
        const char* src = W + ((size_t)s % nst) * STAGE;
        for (int u = 0; u < ppw; ++u) {
            int piece = wv + NW * u;
            glds16(src + piece * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(lds0 + s * STAGE + piece * 1024));
        }
    }
    for (size_t c = 0; c < total; ++c) {
        if (ppw == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const size_t nx = c + 2;
        const char* src = W + (nx % nst) * STAGE;
        const int st = (int)(nx % NSTAGE);
        for (int u = 0; u < ppw; ++u) {
            int piece = wv + NW * u;
            glds16(src + piece * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(lds0 + st * STAGE + piece * 1024));
        }
        // touch the landed stage lightly so the loads are not dead
        acc += *(const float*)(lds + (c % NSTAGE) * STAGE + tid * 4);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 12345.678f) sink[0] = acc;
}

int main(int argc, char** argv) {
    size_t bytes = 512 * 1024;   // one 512x512 bf16 weight matrix
    int iters = 64;
    char* W; float* sink;
    hipMalloc(&W, 8 << 20); hipMemset(W, 1, 8 << 20); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (size_t wb : {(size_t)512 << 10, (size_t)3 << 20}) {
        for (int wgs_per_cu : {1, 2}) {
            for (int nt : {256, 512}) {
                int grid = 256 * wgs_per_cu;
                auto launch = [&]() {
                    if (nt == 256) { hipFuncSetAttribute((const void*)stream_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 32768);
                        stream_kernel<256><<<grid, 256, 3 * 32768 / (wgs_per_cu == 2 ? 1 : 1), 0>>>(W, wb, iters, sink); }
                    else { hipFuncSetAttribute((const void*)stream_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 32768);
                        stream_kernel<512><<<grid, 512, 3 * 32768, 0>>>(W, wb, iters, sink); }
                };
                if (wgs_per_cu == 2) continue;  // 96 KiB LDS per WG: one WG per CU only
                launch(); hipDeviceSynchronize();
                hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                double total = (double)wb * iters * grid;
                printf("panel %zu KiB, %d threads/WG, %d WGs: %.3f ms  -> %.1f GB/s per CU, %.2f TB/s chip\n", wb >> 10, nt, grid, ms,
                       total / grid / (ms * 1e-3) / 1e9, total / (ms * 1e-3) / 1e12);
            }
        }
    }
    return 0;
}
