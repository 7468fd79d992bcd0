// Micro-benchmark: how fast does a CU pull a [M][2048] bf16 matrix's first 1536 columns (dfeat.hip's dH operand: 3 KiB of every 4-KiB
// row) from HBM into LDS by DMA (global_load_lds_dwordx4), as a function of the SHAPE of a 1-KiB piece?
//   P = bytes per row per piece (64: 16 rows x 64 B -- dfeat.hip's K step; 256: 4 rows x 256 B; 1024: one row x 1 KiB)
//   D = pieces in flight per wave, G = workgroups per CU (256 threads each).  No compute: waits + nothing.
//   hipcc --offload-arch=gfx950 -O3 dma_stream.hip -o dma_stream && ./dma_stream
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

__device__ static inline void glds16(const void* src, unsigned lds_wave_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_wave_base) : "memory");
}

// every wave owns ROWS = 32 rows of a 128-row tile and streams their 3 KiB each in pieces of (1024 / P) rows x P bytes, D pieces in
// flight; tiles are handed out round robin (persistent workgroups: tile = blockIdx + k * gridDim)
template <int P, int D>
__global__ __launch_bounds__(256) void k(const char* __restrict__ A, int ntile, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int RPP = 1024 / P;           // rows per piece
    constexpr int SPR = P / 16;             // 16-byte slots per row per piece
    constexpr int NPIECE = 32 / RPP * (3072 / P);   // pieces per wave per tile
    const unsigned base = (unsigned)(uintptr_t)lds + wv * D * 1024;
    const int rp = lane / SPR, ps = lane % SPR;
    int issued = 0;
    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const char* t0 = A + ((size_t)tile * 128 + wv * 32) * 4096;
        // order: all row groups of a K position, then the next K position (the K loop of a GEMM)
        for (int kk = 0; kk < 3072 / P; ++kk)
            for (int g = 0; g < 32 / RPP; ++g) {
                if (issued >= D) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 1) : "memory");
                glds16(t0 + (size_t)(g * RPP + rp) * 4096 + kk * P + ps * 16, __builtin_amdgcn_readfirstlane(base + (issued % D) * 1024));
                ++issued;
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lds[threadIdx.x] == 77 && A == nullptr) sink[0] = 1.f;
}

template <int P, int D> void run(const char* A, int ntile, int wgs_per_cu, float* sink) {
    hipFuncSetAttribute((const void*)k<P, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int lds = 160 * 1024 / wgs_per_cu - 512 > 4 * D * 1024 ? (160 * 1024 / wgs_per_cu - 512) & ~1023 : 4 * D * 1024;   // forces the occupancy
    const int grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<P, D><<<grid, 256, lds>>>(A, ntile, sink);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) k<P, D><<<grid, 256, lds>>>(A, ntile, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)ntile * 128 * 3072;
    printf("piece %4d B/row x %2d rows, %2d in flight per wave, %d WG/CU (%3d KB in flight per CU): %7.1f us  %5.2f TB/s%s\n", P, 1024 / P, D, wgs_per_cu,
           D * 4 * wgs_per_cu, ms / 5 * 1e3, bytes / (ms / 5 * 1e-3) / 1e12, hipGetLastError() == hipSuccess ? "" : "  LAUNCH ERROR");
}

int main() {
    const int ntile = 1200;
    char* A; float* sink;
    hipMalloc(&A, (size_t)ntile * 128 * 4096);
    hipMalloc(&sink, 4);
    hipMemset(A, 1, (size_t)ntile * 128 * 4096);
    for (int wg = 1; wg <= 2; ++wg) {
        run<64, 4>(A, ntile, wg, sink); run<64, 8>(A, ntile, wg, sink); run<64, 16>(A, ntile, wg, sink);
        run<256, 4>(A, ntile, wg, sink); run<256, 8>(A, ntile, wg, sink); run<256, 16>(A, ntile, wg, sink);
        run<1024, 4>(A, ntile, wg, sink); run<1024, 8>(A, ntile, wg, sink); run<1024, 16>(A, ntile, wg, sink);
    }
    run<64, 8>(A, ntile, 4, sink); run<256, 8>(A, ntile, 4, sink); run<1024, 8>(A, ntile, 4, sink);
    return 0;
}
