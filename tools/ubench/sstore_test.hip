// Do scalar stores work on gfx950?  Each wave writes a ballot word with s_store_dwordx4, host checks.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* x, unsigned long long* out) {
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) / 64;
    const float v = x[blockIdx.x * blockDim.x + threadIdx.x];
    const unsigned long long m0 = __builtin_amdgcn_ballot_w64(v > 0.f), m1 = __builtin_amdgcn_ballot_w64(v > 0.5f);
    u4 d = {(unsigned)m0, (unsigned)(m0 >> 32), (unsigned)m1, (unsigned)(m1 >> 32)};
    unsigned long long* p = out + (size_t)w * 2;
    asm volatile("s_store_dwordx4 %0, %1, 0x0" : : "s"(d), "s"(p) : "memory");
    asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}
int main() {
    const int n = 64 * 1024;
    float* hx = new float[n];
    for (int i = 0; i < n; ++i) hx[i] = ((i * 2654435761u) >> 8 & 0xffff) / 65536.0f - 0.3f;
    float* dx; unsigned long long* dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, n / 64 * 16); hipMemset(dout, 0, n / 64 * 16);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dout);
    hipError_t e = hipDeviceSynchronize();
    unsigned long long* ho = new unsigned long long[n / 64 * 2];
    hipMemcpy(ho, dout, n / 64 * 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < n / 64; ++w) {
        unsigned long long m0 = 0, m1 = 0;
        for (int l = 0; l < 64; ++l) { if (hx[w * 64 + l] > 0.f) m0 |= 1ull << l; if (hx[w * 64 + l] > 0.5f) m1 |= 1ull << l; }
        if (ho[2 * w] != m0 || ho[2 * w + 1] != m1) ++bad;
    }
    printf("sync: %s ; %d of %d waves wrong\n", hipGetErrorString(e), bad, n / 64);
    return bad != 0;
}
