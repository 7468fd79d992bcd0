/* Host build of scenerf_amd/csrc/sphere_exact.h for the CPU-side bit-exactness checks (tests/test_sphere_exact.py).
 * TEST INFRASTRUCTURE: compiled by the test with `gcc -O2 -ffp-contract=off -shared -fPIC`; never loaded by the product. */
#include "../../scenerf_amd/csrc/sphere_exact.h"
#include <stddef.h>

void srf_host_acosf(const float* x, float* y, size_t n) {
    for (size_t i = 0; i < n; ++i) y[i] = srf_acosf_u10(x[i]);
}
void srf_host_atan2f(const float* a, const float* b, float* y, size_t n) {
    for (size_t i = 0; i < n; ++i) y[i] = srf_atan2f_u10(a[i], b[i]);
}
/* all floats whose bit pattern is in [lo, hi): count of values where srf_acosf_u10 differs from ref[i - lo] */
size_t srf_host_acosf_range_mismatch(uint32_t lo, uint32_t hi, const float* ref) {
    size_t bad = 0;
    for (uint32_t b = lo; b < hi; ++b) {
        float r = srf_acosf_u10(srf_from_bits(b));
        bad += srf_bits(r) != srf_bits(ref[b - lo]);
    }
    return bad;
}
/* infer-frame points (M,3) -> sphere pixel (M,2) int32 and the float coordinates (M,2) */
void srf_host_points_to_sphere(const float* pts, size_t M, const float* K, const float* iK, const float* consts, int W, int H,
                               int32_t* idx, float* coords, float* pix) {
    srf_sphere_consts sc;
    sc.v_min = consts[0]; sc.v_fov = consts[1]; sc.h_min = consts[2]; sc.h_fov = consts[3]; sc.W = W; sc.H = H;
    for (size_t m = 0; m < M; ++m) {
        float u, v, ox, oy;
        srf_cam_pt_to_pix(K, pts[3 * m], pts[3 * m + 1], pts[3 * m + 2], &u, &v);
        srf_pix_to_sphere_f(iK, sc, u, v, &ox, &oy);
        if (pix) { pix[2 * m] = u; pix[2 * m + 1] = v; }
        if (coords) { coords[2 * m] = ox; coords[2 * m + 1] = oy; }
        idx[2 * m] = srf_round_index(ox);
        idx[2 * m + 1] = srf_round_index(oy);
    }
}
/* pixels (R,2) -> unit directions (R,3) and un-normalised infer-frame view directions (R,3) */
void srf_host_rays(const float* pix, size_t R, const float* iK, const float* T, float* unit, float* viewdir) {
    for (size_t r = 0; r < R; ++r) srf_ray_dir(iK, T, pix[2 * r], pix[2 * r + 1], unit + 3 * r, viewdir + 3 * r);
}
/* unit (R,3), dist (R,S) -> infer-frame sample points (R*S,3) */
void srf_host_sample_points(const float* unit, const float* dist, size_t R, size_t S, const float* T, float* pts) {
    for (size_t r = 0; r < R; ++r)
        for (size_t j = 0; j < S; ++j) srf_sample_point(T, unit + 3 * r, dist[r * S + j], pts + 3 * (r * S + j));
}
