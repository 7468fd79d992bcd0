// What does ds_read_b64_tr_b16 return?  LDS holds a row-major [16][64] u16 matrix with value = 256*row + col.
// Every lane supplies the address of 4 consecutive u16; two lane->address mappings are tried.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned u2v __attribute__((ext_vector_type(2)));
__global__ void k(uint16_t* out, int variant) {
    __shared__ __attribute__((aligned(16))) uint16_t t[16][64];
    for (int i = threadIdx.x; i < 16 * 64; i += 64) t[i / 64][i % 64] = (uint16_t)(256 * (i / 64) + (i % 64));
    __syncthreads();
    const int l = threadIdx.x;
    int row, cq;
    if (variant == 0) { row = (l & 15) >> 2; cq = (l & 3) + 4 * (l >> 4); }       // lanes 4r+c of a 16-group: row r, column quad c
    else { row = l & 3; cq = ((l & 15) >> 2) + 4 * (l >> 4); }                     // lanes r+4c
    const unsigned addr = (unsigned)(uintptr_t)&t[row][4 * cq];
    u2v v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = v[0] & 0xffff; out[l * 4 + 1] = v[0] >> 16; out[l * 4 + 2] = v[1] & 0xffff; out[l * 4 + 3] = v[1] >> 16;
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int variant = 0; variant < 2; ++variant) {
        k<<<1, 64>>>(d, variant);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("variant %d (lane: 4 values as row.col)\n", variant);
        for (int l = 0; l < 64; ++l) {
            printf("  l%02d:", l);
            for (int j = 0; j < 4; ++j) printf(" %d.%02d", h[l * 4 + j] >> 8, h[l * 4 + j] & 255);
            if (l % 4 == 3) printf("\n");
        }
    }
    return 0;
}
