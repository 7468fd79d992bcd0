// Which SIMD does wave w of a workgroup run on?  (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8].)  Printed for the
// workgroup shapes of the kernels here: 256 threads (wide.hip), 512 (fused.hip) and 768 (wgrad.hip), each with a large LDS footprint
// (one workgroup per CU).
// hipcc --offload-arch=gfx950 -O3 simd_map.hip -o simd_map && ./simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    extern __shared__ char lds[];
    const unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}
int main() {
    unsigned* out; hipMalloc((void**)&out, 4 * 16 * 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    for (int threads : {256, 512, 768}) {
        hipMemset(out, 0, 4 * 16 * 8);
        k<<<8, threads, 128 * 1024>>>(out);
        unsigned h[16 * 8];
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        for (int b = 0; b < 2; ++b) {
            printf("%d threads, workgroup %d: simd of wave 0..%d:", threads, b, threads / 64 - 1);
            for (int w = 0; w < threads / 64; ++w) printf(" %u", (h[b * 16 + w] >> 4) & 3);
            printf("   (cu %u, raw %08x)\n", (h[b * 16] >> 8) & 15, h[b * 16]);
        }
    }
    return 0;
}
