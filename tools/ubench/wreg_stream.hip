// Micro-benchmark (feasibility of a fused-MLP variant): every consumer wave streams ITS OWN column slice of the L2-resident weight
// stream straight into a VGPR ring (global_load_dwordx4, D chunks ahead, compiler-counted vmcnt), the activation operand comes from a
// resident LDS buffer, RT x 2 MFMA 32x32x16 per 16-wide K chunk per wave -- no LDS ring, no per-chunk barrier.
//   hipcc --offload-arch=gfx950 -O3 wreg_stream.hip -o wreg_stream && ./wreg_stream
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// Plain loads + a scheduling barrier per step: left to itself hipcc's scheduler sinks the loads of a register ring down to their
// first use (vmcnt(0) in front of every MFMA group, no prefetch at all); inline-asm loads are not an option -- the compiler copies
// an asm output wherever it likes before the data has landed (and reuses the "dead" registers for addresses: memory fault).
__device__ static inline void gload(u32x4& dst, const u32x4* p) { dst = *p; }

// W: nblocks blocks of 16 KiB = [512 rows n][32 B] ; wave w owns rows 64 w .. 64 w + 63 (two 32-row tiles)
template <int D, int RT, bool MFMA>
__global__ __launch_bounds__(512) void k(const u32x4* __restrict__ W, int nblocks, int nchunks, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // resident activations: RT*32 rows x 1 KiB, touched once
    for (int i = tid; i < RT * 32 * 1024 / 16; i += 512) ((u32x4*)lds)[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    __syncthreads();
    const int woff = (wv * 64 + (lane & 31)) * 2 + (lane >> 5);   // uint4 index of this lane's 16 B inside a block (tile 0); tile 1: + 64
    u32x4 ring[D][2];
    f32x16 acc[RT][2];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][j][e] = 0.f;
#pragma unroll
    for (int s = 0; s < D; ++s) {
        const u32x4* b = W + (size_t)(s % nblocks) * 1024;
        gload(ring[s][0], b + woff);
        gload(ring[s][1], b + woff + 64);
    }
    __builtin_amdgcn_sched_barrier(0);
    const int arow = (lane & 31) * 1024;
    int blk = D % nblocks;
    for (int c = 0; c < nchunks; c += D) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
            // activation fragments of this chunk (k slot from the chunk index, XOR swizzle like the real buffer)
            const int kslot = (((c + s) & 31) * 2 + (lane >> 5)) ^ (lane & 15);
            u32x4 a[RT];
#pragma unroll
            for (int r = 0; r < RT; ++r) a[r] = *(const u32x4*)(lds + r * 32768 + arow + (kslot << 4));
            if (MFMA) {
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[r][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ring[s][j]), __builtin_bit_cast(bf16x8, a[r]),
                                                                            acc[r][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[0][j][0] += __builtin_bit_cast(float, ring[s][j].x) + __builtin_bit_cast(float, a[0].x);
            }
            const u32x4* b = W + (size_t)blk * 1024;
            gload(ring[s][0], b + woff);
            gload(ring[s][1], b + woff + 64);
            blk = blk + 1 == nblocks ? 0 : blk + 1;
            __builtin_amdgcn_sched_barrier(0);   // nothing moves across a step boundary
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) t += acc[r][j][e];
    if (t == 12345.678f) sink[0] = t;
}

template <int D, int RT, bool MFMA>
void run(const u32x4* W, float* sink, int wgs) {
    const int nblocks = 858, nchunks = 288 * 8 / D * D;   // ~ a forward pass worth of chunks per workgroup
    hipFuncSetAttribute((const void*)k<D, RT, MFMA>, hipFuncAttributeMaxDynamicSharedMemorySize, RT * 32768);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<D, RT, MFMA><<<wgs, 512, RT * 32768>>>(W, nblocks, nchunks, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<D, RT, MFMA><<<wgs, 512, RT * 32768>>>(W, nblocks, nchunks, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double rounds = (wgs + 255) / 256;
    const double us_chunk = ms * 1e3 / rounds / nchunks;
    const double flops = 2.0 * (RT * 32) * 512 * 16 * (double)nchunks * wgs;
    printf("D=%d rows=%3d %s  %4d WGs: %.3f ms  %.3f us/chunk (%.0f cycles @2.0GHz)  W stream %.1f GB/s/CU  %s%.0f TF/s\n", D, RT * 32,
           MFMA ? "mfma" : "load", wgs, ms, us_chunk, us_chunk * 2000, 16384.0 / us_chunk / 1e3, MFMA ? "" : "(no mfma) ", MFMA ? flops / ms / 1e9 : 0.0);
}

int main() {
    u32x4* W; float* sink;
    hipMalloc(&W, 858 * 16384); hipMemset(W, 0, 858 * 16384); hipMalloc(&sink, 4);
    run<4, 3, false>(W, sink, 256);
    run<8, 3, false>(W, sink, 256);
    run<4, 3, true>(W, sink, 256);
    run<6, 3, true>(W, sink, 256);
    run<8, 3, true>(W, sink, 256);
    run<4, 4, true>(W, sink, 256);
    run<6, 4, true>(W, sink, 256);
    run<4, 2, true>(W, sink, 256);
    run<8, 2, true>(W, sink, 256);
    run<6, 3, true>(W, sink, 1600);
    return 0;
}
