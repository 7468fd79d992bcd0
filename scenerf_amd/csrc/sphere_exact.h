// The operation sequence torch-CPU performs between a pixel and its rounded sphere pixel, restated so that the sphere index is a
// function of the inputs and of nothing else (reference: scenerf/models/spherical_mapping.py:80-115, utils.py:177-182, 298-315).
//
// The reference's sphere pixel is `round((angle - min) / fov * (size - 1))` of an angle that went through torch.acos / torch.atan2.  A
// sample within an ulp of a .5 boundary lands on the neighbouring texel when the last bit of anything upstream differs, so "each op
// within 1 ulp" is not enough for SURVEY section 8d's "sphere indices bit-exact": 13-15 of 158,400 samples differed in round 4 (ocml's
// acosf / atan2f, and a 2-norm summed in another order).  What torch-CPU does, established on the build container and held by
// tests/test_sphere_exact.py without a GPU:
//   * `A @ x` for the 3x3 / 4x4 matrices: k-ordered fma chains (srf_dot3 / srf_dot4) -- equal to torch on every element tried;
//   * torch.linalg.norm / F.normalize over 3 elements: sqrt(fma(z, z, fma(y, y, x * x))) -- equal on every element;
//   * torch.atan2 (float32): SLEEF's Sleef_atan2f{8,16}_u10, the same bits with AVX2 and AVX-512 -- srf_atan2f_u10 below is that
//     routine's operation sequence (double-float arithmetic, FMA forms), equal to torch.atan2 on 2^24 random pairs and on the edge cases;
//   * torch.acos (float32): NOT one routine.  With MKL it is VML's vmsAcos(VML_HA), closed source, and its last bit depends on the
//     instruction set MKL dispatches to (AVX-512 vs AVX2 kernels: 207 of 4,194,304 inputs differ; tools/sleef_check/acos_isa_probe.py);
//     without MKL it is SLEEF's Sleef_acosf*_u10.  The rule pinned here is SLEEF u10 -- the open routine torch
//     itself ships, ISA-stable -- and srf_acosf_u10 equals torch's own build of it on every float in [-1, 1] (exhaustive check).  Against
//     a reference run whose acos was MKL's AVX-512 kernel, 4-6 samples per million land on the neighbouring row (never column).
// The two SLEEF sequences were written from the published algorithm (sleefsimdsp.c, Boost licence) and compared instruction by instruction
// with the routines in torch's libtorch_cpu.so (tools/sleef_check/sleef_dis.py).  Every multiply-add below is explicit: a translation
// unit that includes this header must be compiled with -ffp-contract=off (rays.hip is; so is the host-side check).
//
// One source, two compilers: hipcc (device, used by rays.hip) and gcc (host, used ONLY by tests/ to compare this sequence with torch,
// torch's SLEEF and the oracle bit for bit without a GPU).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SRF_X __device__ static inline
#define SRF_FMA(a, b, c) __builtin_fmaf((a), (b), (c))
#define SRF_SQRT(a) __builtin_sqrtf(a)          // correctly rounded (hipcc default: -fhip-fp32-correctly-rounded-divide-sqrt)
#define SRF_RINT(a) __builtin_rintf(a)
#else
#include <math.h>
#define SRF_X static inline
#define SRF_FMA(a, b, c) fmaf((a), (b), (c))
#define SRF_SQRT(a) sqrtf(a)
#define SRF_RINT(a) rintf(a)
#endif

typedef struct { float x, y; } srf_f2;          // an unevaluated sum x + y, |y| <= ulp(x)/2

SRF_X uint32_t srf_bits(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
SRF_X float srf_from_bits(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }
SRF_X float srf_mulsign(float x, float s) { return srf_from_bits(srf_bits(x) ^ (srf_bits(s) & 0x80000000u)); }
SRF_X float srf_fabs(float x) { return srf_from_bits(srf_bits(x) & 0x7fffffffu); }

// ---- double-float helpers (the FMA forms)
SRF_X srf_f2 srf_df(float x, float y) { srf_f2 r; r.x = x; r.y = y; return r; }
SRF_X srf_f2 srf_dfadd_f_f(float x, float y) {              // |x| >= |y|
    float s = x + y;
    return srf_df(s, (x - s) + y);
}
SRF_X srf_f2 srf_dfadd_f2_f(srf_f2 x, float y) {            // |x| >= |y|
    float s = x.x + y;
    return srf_df(s, ((x.x - s) + y) + x.y);
}
SRF_X srf_f2 srf_dfadd_f_f2(float x, srf_f2 y) {            // |x| >= |y|
    float s = x + y.x;
    return srf_df(s, ((x - s) + y.x) + y.y);
}
SRF_X srf_f2 srf_dfadd_f2_f2(srf_f2 x, srf_f2 y) {          // |x| >= |y|
    float s = x.x + y.x;
    return srf_df(s, (((x.x - s) + y.x) + x.y) + y.y);
}
SRF_X srf_f2 srf_dfadd2_f_f2(float x, srf_f2 y) {
    float s = x + y.x;
    float v = s - x;
    return srf_df(s, ((x - (s - v)) + (y.x - v)) + y.y);
}
SRF_X srf_f2 srf_dfsub_f2_f2(srf_f2 x, srf_f2 y) {          // |x| >= |y|
    float s = x.x - y.x;
    float t = x.x - s;
    t = t - y.x;
    t = t + x.y;
    return srf_df(s, t - y.y);
}
SRF_X srf_f2 srf_dfmul_f_f(float x, float y) {
    float s = x * y;
    return srf_df(s, SRF_FMA(x, y, -s));
}
SRF_X srf_f2 srf_dfmul_f2_f(srf_f2 x, float y) {
    float s = x.x * y;
    return srf_df(s, SRF_FMA(x.y, y, SRF_FMA(x.x, y, -s)));
}
SRF_X srf_f2 srf_dfmul_f2_f2(srf_f2 x, srf_f2 y) {
    float s = x.x * y.x;
    return srf_df(s, SRF_FMA(x.x, y.y, SRF_FMA(x.y, y.x, SRF_FMA(x.x, y.x, -s))));
}
SRF_X srf_f2 srf_dfsqu_f2(srf_f2 x) {
    float s = x.x * x.x;
    return srf_df(s, SRF_FMA(x.x + x.x, x.y, SRF_FMA(x.x, x.x, -s)));
}
SRF_X srf_f2 srf_dfrec_f(float d) {
    float s = 1.0f / d;
    return srf_df(s, s * SRF_FMA(-d, s, 1.0f));
}
SRF_X srf_f2 srf_dfdiv_f2_f2(srf_f2 n, srf_f2 d) {
    float t = 1.0f / d.x;
    float s = n.x * t;
    float u = SRF_FMA(t, n.x, -s);
    float v = SRF_FMA(-d.y, t, SRF_FMA(-d.x, t, 1.0f));
    return srf_df(s, SRF_FMA(s, v, SRF_FMA(n.y, t, u)));
}
SRF_X srf_f2 srf_dfsqrt_f(float d) {
    float t = SRF_SQRT(d);
    srf_f2 r = srf_dfmul_f2_f2(srf_dfadd2_f_f2(d, srf_dfmul_f_f(t, t)), srf_dfrec_f(t));
    return srf_df(r.x * 0.5f, r.y * 0.5f);
}

// ---- SLEEF xacosf_u1 (torch.acos on a CPU build without MKL; the pinned rule)
SRF_X float srf_acosf_u10(float d) {
    const float ad = srf_fabs(d);
    const int o = ad < 0.5f;
    const float x2 = o ? d * d : (1.0f - ad) * 0.5f;
    srf_f2 x = o ? srf_df(ad, 0.0f) : srf_dfsqrt_f(x2);
    if (ad == 1.0f) x = srf_df(0.0f, 0.0f);
    float u = +0.4197454825e-1f;
    u = SRF_FMA(u, x2, +0.2424046025e-1f);
    u = SRF_FMA(u, x2, +0.4547423869e-1f);
    u = SRF_FMA(u, x2, +0.7495029271e-1f);
    u = SRF_FMA(u, x2, +0.1666677296e+0f);
    u = (x.x * x2) * u;
    srf_f2 y = srf_dfsub_f2_f2(srf_df(3.1415927410125732422f / 2, -8.7422776573475857731e-08f / 2),
                               srf_dfadd_f_f(srf_mulsign(x.x, d), srf_mulsign(u, d)));
    x = srf_dfadd_f2_f(x, u);
    if (!o) y = srf_df(x.x * 2.0f, x.y * 2.0f);
    if (!o && d < 0.0f) y = srf_dfsub_f2_f2(srf_df(3.1415927410125732422f, -8.7422776573475857731e-08f), y);
    return y.x + y.y;
}

// ---- torch.atan2 on the CPU for float32 = SLEEF xatan2f_u1
SRF_X srf_f2 srf_atan2kf_u1(srf_f2 y, srf_f2 x) {
    int q = x.x < 0.0f ? -2 : 0;
    if (x.x < 0.0f) { x.x = -x.x; x.y = -x.y; }
    const int p = x.x < y.x;
    if (p) q += 1;
    srf_f2 s = p ? srf_df(-x.x, -x.y) : y;
    srf_f2 t = p ? y : x;
    s = srf_dfdiv_f2_f2(s, t);
    t = srf_dfsqu_f2(s);
    { float n = t.x + t.y; t = srf_df(n, (t.x - n) + t.y); }          // normalise
    float u = -0.00176397908944636583328247f;
    u = SRF_FMA(u, t.x, 0.0107900900766253471374512f);
    u = SRF_FMA(u, t.x, -0.0309564601629972457885742f);
    u = SRF_FMA(u, t.x, 0.0577365085482597351074219f);
    u = SRF_FMA(u, t.x, -0.0838950723409652709960938f);
    u = SRF_FMA(u, t.x, 0.109463557600975036621094f);
    u = SRF_FMA(u, t.x, -0.142626821994781494140625f);
    u = SRF_FMA(u, t.x, 0.199983194470405578613281f);
    t = srf_dfmul_f2_f2(t, srf_dfadd_f_f(-0.333332866430282592773438f, u * t.x));
    t = srf_dfmul_f2_f2(s, srf_dfadd_f_f2(1.0f, t));
    return srf_dfadd_f2_f2(srf_dfmul_f2_f(srf_df(1.5707963705062866211f, -4.3711388286737928865e-08f), (float)q), t);
}
SRF_X float srf_atan2f_u10(float y, float x) {
    if (srf_fabs(x) < 2.9387372783541830947e-39f) { x *= 16777216.0f; y *= 16777216.0f; }
    const srf_f2 d = srf_atan2kf_u1(srf_df(srf_fabs(y), 0.0f), srf_df(x, 0.0f));
    float r = d.x + d.y;
    const float inf = srf_from_bits(0x7f800000u);
    const int xinf = srf_fabs(x) == inf;
    const float HPI = 1.5707963705062866211f, QPI = 0.78539818525314331055f, PI = 3.1415927410125732422f;
    r = srf_mulsign(r, x);
    if (xinf || x == 0.0f) r = HPI - (xinf ? srf_mulsign(HPI, x) : 0.0f);
    if (srf_fabs(y) == inf) r = HPI - (xinf ? srf_mulsign(QPI, x) : 0.0f);
    if (y == 0.0f) r = (srf_bits(x) >> 31) ? PI : 0.0f;
    if (x != x || y != y) return srf_from_bits(0xffffffffu);
    return srf_mulsign(r, y);
}

// ---- the chain: infer-frame point -> (pixel) -> unit-depth camera point -> angles -> sphere pixel
// k-ordered fma chain == what the BLAS sgemm micro-kernel behind `K @ p` does for a length-3 / length-4 dot product
SRF_X float srf_dot3(float a0, float a1, float a2, float x, float y, float z) { return SRF_FMA(a2, z, SRF_FMA(a1, y, a0 * x)); }
SRF_X float srf_dot4(float a0, float a1, float a2, float a3, float x, float y, float z, float w) {
    return SRF_FMA(a3, w, SRF_FMA(a2, z, SRF_FMA(a1, y, a0 * x)));
}
// compute_direction_from_pixels + the un-normalised infer-frame view direction (utils.py:177-182, 131-135, 170): d = inv_K @ [u, v, 1];
// unit = F.normalize(d) = d / max(||d||, 1e-12) with torch's 2-norm (sequential acc + x*x over the 3 elements, contracted to fma by its
// compiler: checked against torch.linalg.norm / F.normalize bit for bit); viewdir = T[:3,:3] @ d
SRF_X float srf_norm3(float x, float y, float z) { return SRF_SQRT(SRF_FMA(z, z, SRF_FMA(y, y, x * x))); }
SRF_X void srf_ray_dir(const float* iK, const float* T, float u, float v, float* unit, float* viewdir) {
    const float dx = srf_dot3(iK[0], iK[1], iK[2], u, v, 1.0f);
    const float dy = srf_dot3(iK[3], iK[4], iK[5], u, v, 1.0f);
    const float dz = srf_dot3(iK[6], iK[7], iK[8], u, v, 1.0f);
    float n = srf_norm3(dx, dy, dz);
    n = n < 1e-12f ? 1e-12f : n;               // clamp_min(eps)
    unit[0] = dx / n; unit[1] = dy / n; unit[2] = dz / n;
    viewdir[0] = srf_dot3(T[0], T[1], T[2], dx, dy, dz);
    viewdir[1] = srf_dot3(T[4], T[5], T[6], dx, dy, dz);
    viewdir[2] = srf_dot3(T[8], T[9], T[10], dx, dy, dz);
}
// sample point in the infer frame: (dist * unit) then T @ [p, 1] (utils.py:87 / 217, 161-166)
SRF_X void srf_sample_point(const float* T, const float* unit, float d, float* q) {
    const float px = d * unit[0], py = d * unit[1], pz = d * unit[2];
    q[0] = srf_dot4(T[0], T[1], T[2], T[3], px, py, pz, 1.0f);
    q[1] = srf_dot4(T[4], T[5], T[6], T[7], px, py, pz, 1.0f);
    q[2] = srf_dot4(T[8], T[9], T[10], T[11], px, py, pz, 1.0f);
}
typedef struct {
    float v_min, v_fov, h_min, h_fov;
    int W, H;
} srf_sphere_consts;

// cam_pts_2_pix (utils.py:298-315): K @ p, perspective divide where z > 0 else (-1, -1)
SRF_X void srf_cam_pt_to_pix(const float* K, float qx, float qy, float qz, float* u, float* v) {
    const float h0 = srf_dot3(K[0], K[1], K[2], qx, qy, qz);
    const float h1 = srf_dot3(K[3], K[4], K[5], qx, qy, qz);
    const float h2 = srf_dot3(K[6], K[7], K[8], qx, qy, qz);
    *u = -1.0f; *v = -1.0f;
    if (h2 > 0.0f) { *u = h0 / h2; *v = h1 / h2; }
}
// from_pixels at depth 1 + cam_pts_2_sphere_coords (spherical_mapping.py:80-115), the float coordinates before round()
SRF_X void srf_pix_to_sphere_f(const float* iK, srf_sphere_consts sc, float u, float v, float* ox, float* oy) {
    const float PI_F32 = 3.14159265358979323846f;
    // pix_2_cam_pts: inv_K @ [u, v, 1] (then * depth 1: exact)
    const float cx = srf_dot3(iK[0], iK[1], iK[2], u, v, 1.0f);
    const float cy = srf_dot3(iK[3], iK[4], iK[5], u, v, 1.0f);
    const float cz = srf_dot3(iK[6], iK[7], iK[8], u, v, 1.0f);
    // torch.linalg.norm(ord=2, dim=1) over 3 contiguous floats: sequential acc + x*x in float, the compiler's contraction -> fma
    const float cd = srf_norm3(cx, cy, cz);
    const float v_angle = srf_acosf_u10(-cy / cd) / PI_F32 * 180.0f;
    const float h_angle = 180.0f - srf_atan2f_u10(cz, cx) / PI_F32 * 180.0f;
    *ox = (h_angle - sc.h_min) / sc.h_fov * (float)(sc.W - 1);
    *oy = (v_angle - sc.v_min) / sc.v_fov * (float)(sc.H - 1);
}
// torch.round(...).long(), with the conversion made defined for far-out / NaN coordinates (they are out of every map either way)
SRF_X int32_t srf_round_index(float o) {
    float r = SRF_RINT(o);                     // half to even, like torch.round
    r = r < -1.0e9f ? -1.0e9f : r;
    r = r > 1.0e9f ? 1.0e9f : r;
    if (!(r == r)) r = -1.0e9f;
    return (int32_t)r;
}
