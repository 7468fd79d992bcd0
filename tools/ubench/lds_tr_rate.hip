// LDS read throughput per CU for the access shapes of the weight-gradient kernel: ds_read_b64_tr_b16 (transposing, 8 B per lane)
// against ds_read_b64 and ds_read_b128, 8 waves per CU reading with the kernel's address pattern (rows 512 B apart, 64-B chunk
// index XOR (row & 3)).  Prints bytes per cycle per CU.
// hipcc --offload-arch=gfx950 -O3 lds_tr_rate.hip -o lds_tr_rate && ./lds_tr_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(int iters, unsigned long long* out) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int g = lane >> 4, r = (lane >> 2) & 3, cq = lane & 3;
    unsigned base = (unsigned)(uintptr_t)lds + (8 * (g >> 1) + r) * 512 + (16 * (g & 1) + 4 * cq) * 2;
    unsigned ad[6];
    for (int i = 0; i < 6; ++i) ad[i] = base + ((((wv & 7) + i) & 7 ^ r) << 6);
    if (MODE == 2) for (int i = 0; i < 6; ++i) ad[i] = (unsigned)(uintptr_t)lds + lane * 16 + i * 1024 + wv * 8192;   // plain contiguous b128
    unsigned acc = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const unsigned so = (it & 3) * 32768;
        if (MODE == 0) {
            u32x2 v[12];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v[2 * i]) : "v"(ad[i] + so) : "memory");
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(v[2 * i + 1]) : "v"(ad[i] + so) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                         "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]));
#pragma unroll
            for (int i = 0; i < 12; ++i) acc ^= v[i][0] ^ v[i][1];
        } else if (MODE == 1) {
            u32x2 v[12];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                asm volatile("ds_read_b64 %0, %1" : "=v"(v[2 * i]) : "v"(ad[i] + so) : "memory");
                asm volatile("ds_read_b64 %0, %1 offset:2048" : "=v"(v[2 * i + 1]) : "v"(ad[i] + so) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                         "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]));
#pragma unroll
            for (int i = 0; i < 12; ++i) acc ^= v[i][0] ^ v[i][1];
        } else {
            u32x4 v[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(v[i]) : "v"(ad[i] + (so & 0)) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]));
#pragma unroll
            for (int i = 0; i < 6; ++i) acc ^= v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 0x12345678u) out[1000] = acc;
}
template <int MODE> void run(const char* name, int waves) {
    unsigned long long* out; hipMalloc(&out, 8 * 2048);
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    const int iters = 4000, wgs = 256;
    k<MODE><<<wgs, waves * 64, 128 * 1024>>>(iters, out);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<wgs, waves * 64, 128 * 1024>>>(iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipDeviceSynchronize();
    unsigned long long h[256]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0; for (int i = 0; i < 256; ++i) cyc += h[i]; cyc /= 256;
    const double bytes = (double)iters * waves * 64 * 96;   // 96 B per lane per iteration in every mode
    printf("%-22s %d waves/CU: %.0f cycles (s_memtime/readcyclecounter units) for %d iterations -> %.1f B per counter tick per CU\n", name, waves, cyc, iters, bytes / cyc);
    printf("    %.3f ms -> %.1f GB/s per CU, %.1f B per cycle at 2.4 GHz\n", ms, bytes / ms / 1e6, bytes / ms / 1e6 / 2.4);
    hipFree(out);
}
int main() {
    for (int w : {4, 8}) {
        if (w == 4) { run<0>("ds_read_b64_tr_b16", 4); run<1>("ds_read_b64", 4); run<2>("ds_read_b128", 4); }
        else { run<0>("ds_read_b64_tr_b16", 8); run<1>("ds_read_b64", 8); run<2>("ds_read_b128", 8); }
    }
    return 0;
}
