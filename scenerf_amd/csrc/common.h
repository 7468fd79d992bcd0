// Shared helpers for libscenerf_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/scenerf_hip.h"

#define WAVE 64
// L2 warm-up of the 128-row fused kernels' first dispatch round (wide.hip; scenerf_hip_test_set_tuning overrides)
#ifndef SRF_WARM_WIDE_DEFAULT
#define SRF_WARM_WIDE_DEFAULT 1
#endif
// microseconds the feature-gradient launch is held back behind the batched weight-gradient launch it runs beside (mlp.hip: srf_delay_kernel)
#ifndef SRF_DFEAT_DELAY_US_DEFAULT
#define SRF_DFEAT_DELAY_US_DEFAULT 12
#endif
int srf_dfeat_delay_us();

// ---- error plumbing: never abort, report through the C ABI ---------------------------------------------
void srf_set_error(const char* fmt, ...);
#define SRF_CHECK(cond, ...)                                                             \
    do {                                                                                 \
        if (!(cond)) { srf_set_error(__VA_ARGS__); return 1; }                           \
    } while (0)
#define SRF_HIP(expr)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            srf_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return 2;                                                                    \
        }                                                                                \
    } while (0)

// ---- launch + optional per-kernel timing ------------------------------------------------------------------
bool srf_prof_on();
void srf_prof_begin(hipStream_t s, const char* name, double flops, double bytes);
void srf_prof_end(hipStream_t s);

struct SrfLaunchScope {
    hipStream_t s;
    bool on;
    SrfLaunchScope(hipStream_t s_, const char* name, double flops = 0, double bytes = 0) : s(s_), on(srf_prof_on()) {
        if (on) srf_prof_begin(s, name, flops, bytes);
    }
    ~SrfLaunchScope() {
        if (on) srf_prof_end(s);
    }
};
#define SRF_LAUNCH_CHECK(name)                                                           \
    do {                                                                                 \
        hipError_t e_ = hipGetLastError();                                               \
        if (e_ != hipSuccess) { srf_set_error("launch %s: %s", name, hipGetErrorString(e_)); return 3; } \
    } while (0)

// ---- per-device one-time state ------------------------------------------------------------------------------
// A process may drive several GPUs (and several host threads): hipFuncSetAttribute is a per-device attribute and device
// allocations belong to the device that was current when they were made, so "once" means once per device ordinal, under a lock.
#include <mutex>
#define SRF_MAX_DEVICES 64
int srf_device();                 // ordinal of the calling thread's current device (hipGetDevice), -1 on error
const char* srf_zero_page();      // 4 KiB of device zeros on the current device (allocated on first use), nullptr on error
struct SrfPerDeviceOnce {
    std::mutex mu;
    bool done[SRF_MAX_DEVICES] = {};
};
#define SRF_ONCE_PER_DEVICE(...)                                                          \
    do {                                                                                  \
        static SrfPerDeviceOnce once_;                                                    \
        const int dev_ = srf_device();                                                    \
        SRF_CHECK(dev_ >= 0 && dev_ < SRF_MAX_DEVICES, "no current HIP device");         \
        std::lock_guard<std::mutex> lk_(once_.mu);                                        \
        if (!once_.done[dev_]) {                                                          \
            __VA_ARGS__;                                                                  \
            once_.done[dev_] = true;                                                      \
        }                                                                                 \
    } while (0)

// rows from which the fused ResnetFC kernels run (scenerf_cfg.fused_min_rows: 0 = default, < 0 = never)
static inline bool srf_use_fused(const scenerf_cfg* cfg, int M) {
    const int th = cfg->fused_min_rows == 0 ? SCENERF_FUSED_MIN_ROWS_DEFAULT : cfg->fused_min_rows;
    return th > 0 && M >= th;
}

// row blocks from which the 128-row fused kernels (wide.hip) replace the 64-row ones when selected: three quarters of the CUs busy
#define SRF_WIDE_MIN_BLOCKS 192
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline hipStream_t as_stream(scenerf_stream_t s) { return (hipStream_t)s; }

// ---- bf16 bit helpers (round-to-nearest-even, like torch's .to(bfloat16)) -----------------------------------
typedef uint16_t bf16_t;
__host__ __device__ static inline float bf16_to_f32(bf16_t h) {
    union { uint32_t u; float f; } v;
    v.u = ((uint32_t)h) << 16;
    return v.f;
}
__host__ __device__ static inline bf16_t f32_to_bf16(float f) {
    union { uint32_t u; float f; } v;
    v.f = f;
    uint32_t u = v.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // quiet NaN
    uint32_t r = 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)((u + r) >> 16);
}
// two floats -> packed bf16 pair, round-to-nearest-even: ONE v_cvt_pk_bf16_f32 on gfx950 (same rounding as f32_to_bf16)
typedef __attribute__((ext_vector_type(2))) __bf16 srf_bf16x2;
typedef __attribute__((ext_vector_type(2))) float srf_f32x2;
__device__ static inline uint32_t pack_bf16x2(float lo, float hi) {
    srf_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, srf_bf16x2));
}
__device__ static inline float bf16lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ static inline float bf16hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
// relu on two packed bf16 in ONE instruction: as signed 16-bit integers every negative bf16 (sign bit set) is a
// negative int16 and every non-negative bf16 keeps its order, so max(x, 0) per half is v_pk_max_i16 (NaNs with the
// sign bit set become 0, like fmaxf(x, 0)).
typedef short srf_short2 __attribute__((ext_vector_type(2)));
__device__ static inline uint32_t relu_bf16x2(uint32_t w) {
    srf_short2 v = __builtin_bit_cast(srf_short2, w);
    srf_short2 z = {0, 0};
    v = __builtin_elementwise_max(v, z);
    return __builtin_bit_cast(uint32_t, v);
}

template <typename T> struct ActIO;
template <> struct ActIO<float> {
    __device__ static inline float ld(const void* p, size_t i) { return ((const float*)p)[i]; }
    __device__ static inline void st(void* p, size_t i, float v) { ((float*)p)[i] = v; }
};
template <> struct ActIO<bf16_t> {
    __device__ static inline float ld(const void* p, size_t i) { return bf16_to_f32(((const bf16_t*)p)[i]); }
    __device__ static inline void st(void* p, size_t i, float v) { ((bf16_t*)p)[i] = f32_to_bf16(v); }
};

// 8 consecutive activations <-> 8 floats (16 B for bf16, 32 B for fp32)
template <typename T> __device__ static inline void load8(const void* base, size_t idx, float* v);
template <> __device__ inline void load8<bf16_t>(const void* base, size_t idx, float* v) {
    uint4 t = *(const uint4*)((const bf16_t*)base + idx);
    v[0] = bf16lo(t.x); v[1] = bf16hi(t.x); v[2] = bf16lo(t.y); v[3] = bf16hi(t.y);
    v[4] = bf16lo(t.z); v[5] = bf16hi(t.z); v[6] = bf16lo(t.w); v[7] = bf16hi(t.w);
}
template <> __device__ inline void load8<float>(const void* base, size_t idx, float* v) {
    float4 a = *(const float4*)((const float*)base + idx), b = *(const float4*)((const float*)base + idx + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <typename T> __device__ static inline void store8(void* base, size_t idx, const float* v);
template <> __device__ inline void store8<bf16_t>(void* base, size_t idx, const float* v) {
    *(uint4*)((bf16_t*)base + idx) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                                pack_bf16x2(v[6], v[7]));
}
template <> __device__ inline void store8<float>(void* base, size_t idx, const float* v) {
    *(float4*)((float*)base + idx) = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)((float*)base + idx + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// wave-level helpers (64 lanes)
// ---- counter-based random numbers made inside kernels (Philox4x32-10, the generator behind torch's device RNG): keyed by a 64-bit seed,
// counted by (element, call).  Used where a torch.rand / torch.randn call in front of a kernel would only feed that kernel: a launch of
// its own on a step's critical chain, plus two generator-state fills at the top of every hipGraph replay.
__device__ static inline void srf_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint64_t seed, uint32_t (&out)[4]) {
    uint32_t c[4] = {c0, c1, c2, 0u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll 1
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}
__device__ static inline float srf_u01(uint32_t x) { return (float)(x >> 8) * (1.f / 16777216.f); }                  // [0, 1), 24 bits (torch.rand)
__device__ static inline float srf_normal(uint32_t a, uint32_t b) {                                                  // Box-Muller on (0, 1] x [0, 1)
    const float u1 = ((float)(a >> 8) + 1.f) * (1.f / 16777216.f), u2 = srf_u01(b);
    return sqrtf(-2.f * logf(u1)) * cosf(6.283185307179586f * u2);
}

__device__ static inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
