// What does one s_barrier per loop iteration cost for a 512-thread workgroup (1 per CU, big LDS footprint)?
// hipcc --offload-arch=gfx950 -O3 barrier_cost.hip -o barrier_cost && ./barrier_cost
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(512) void k(int iters, int* out) {
    extern __shared__ char lds[];
    int acc = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        if (MODE >= 1) __builtin_amdgcn_s_barrier();
        if (MODE >= 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        if (MODE >= 3) { acc += ((int*)lds)[(threadIdx.x * 4 + i) & 1023]; }
        asm volatile("" : "+v"(acc));
    }
    if (acc == 0x7fffffff) out[0] = acc;
}
template <int MODE> void run(const char* name) {
    int* out; hipMalloc(&out, 4);
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 310, wgs = 2400;
    k<MODE><<<wgs, 512, 156 * 1024>>>(iters, out);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<MODE><<<wgs, 512, 156 * 1024>>>(iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-28s %.3f ms per launch -> %.1f ns per iteration per WG-round (10 rounds)\n", name, ms, ms * 1e6 / 10 / iters);
}
int main() {
    run<0>("empty loop");
    run<1>("s_barrier");
    run<2>("s_barrier + vmcnt(0)");
    run<3>("s_barrier + ds_read");
    return 0;
}
