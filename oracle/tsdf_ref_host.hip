// TEST INFRASTRUCTURE ONLY (oracle/): host harness around the REFERENCE's own TSDF kernel.
//
// The kernel text is the pycuda SourceModule string of /root/reference/scenerf/data/utils/fusion.py:72-145.  It is plain CUDA C, so
// hipcc compiles it verbatim for gfx950; oracle/build_ref.py extracts it from the mounted reference at build time into
// oracle/_ref/tsdf_ref_kernel.inc (git-ignored: reference sources are never committed) and compiles this file into
// oracle/_ref/libtsdf_ref.so.  This harness is what pycuda does around the kernel (fusion.py:222-244): every small array is copied
// in per call (cuda.InOut), the three volumes live on the device, one launch per "gpu loop".  It exists to mint golden volumes on the
// GPU box (tests/golden/make_golden_tsdf_gpu.py) that pin oracle/tsdf_oracle.py::integrate_gpu_semantics and the product kernel
// (scenerf_amd/csrc/tsdf.hip) bit for bit.  The product never links or loads it.
//
// One deviation from pycuda's allocation, none from the arithmetic: the reference kernel's bound check lets voxel_idx == n through
// (`voxel_idx > n`), one element past each volume; the device volumes here carry one element of padding so that this access is
// defined.  That element is not part of any result.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#include "_ref/tsdf_ref_kernel.inc"

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            std::snprintf(g_err, sizeof g_err, "%s: %s", #x, hipGetErrorString(e_));           \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

static char g_err[512];

struct Vol {
    float *tsdf, *weight, *color;
    long long n;
};

extern "C" const char* tsdf_ref_last_error() { return g_err; }

extern "C" int tsdf_ref_create(const float* tsdf, const float* weight, const float* color, long long n, void** out) {
    Vol* v = new Vol{nullptr, nullptr, nullptr, n};
    CK(hipMalloc(&v->tsdf, (n + 1) * sizeof(float)));
    CK(hipMalloc(&v->weight, (n + 1) * sizeof(float)));
    CK(hipMalloc(&v->color, (n + 1) * sizeof(float)));
    CK(hipMemset(v->tsdf, 0, (n + 1) * sizeof(float)));
    CK(hipMemset(v->weight, 0, (n + 1) * sizeof(float)));
    CK(hipMemset(v->color, 0, (n + 1) * sizeof(float)));
    CK(hipMemcpy(v->tsdf, tsdf, n * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemcpy(v->weight, weight, n * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemcpy(v->color, color, n * sizeof(float), hipMemcpyHostToDevice));
    *out = v;
    return 0;
}

// fusion.py:222-244: one launch per gpu_loop_idx with (vol_dim, vol_origin, cam_intr, cam_pose, other_params, color_im, depth_im) as
// float arrays; block = (threads, 1, 1), grid = (gx, gy, gz)
extern "C" int tsdf_ref_integrate(void* vol, const float vol_dim[3], const float vol_origin[3], const float cam_intr[9],
                                  const float cam_pose[16], float voxel_size, int im_h, int im_w, float trunc_margin, float obs_weight,
                                  const float* color_im, const float* depth_im, int threads, int gx, int gy, int gz, int n_loops) {
    Vol* v = (Vol*)vol;
    float *d_dim, *d_org, *d_K, *d_P, *d_other, *d_col, *d_dep;
    const size_t npx = (size_t)im_h * im_w;
    CK(hipMalloc(&d_dim, 3 * 4)); CK(hipMalloc(&d_org, 3 * 4)); CK(hipMalloc(&d_K, 9 * 4)); CK(hipMalloc(&d_P, 16 * 4));
    CK(hipMalloc(&d_other, 6 * 4)); CK(hipMalloc(&d_col, npx * 4)); CK(hipMalloc(&d_dep, npx * 4));
    CK(hipMemcpy(d_dim, vol_dim, 12, hipMemcpyHostToDevice)); CK(hipMemcpy(d_org, vol_origin, 12, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_K, cam_intr, 36, hipMemcpyHostToDevice)); CK(hipMemcpy(d_P, cam_pose, 64, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_col, color_im, npx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_dep, depth_im, npx * 4, hipMemcpyHostToDevice));
    for (int loop = 0; loop < n_loops; ++loop) {
        const float other[6] = {(float)loop, voxel_size, (float)im_h, (float)im_w, trunc_margin, obs_weight};
        CK(hipMemcpy(d_other, other, 24, hipMemcpyHostToDevice));
        integrate<<<dim3(gx, gy, gz), dim3(threads, 1, 1)>>>(v->tsdf, v->weight, v->color, d_dim, d_org, d_K, d_P, d_other, d_col, d_dep);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
    }
    for (float* p : {d_dim, d_org, d_K, d_P, d_other, d_col, d_dep}) (void)hipFree(p);
    return 0;
}

extern "C" int tsdf_ref_read(void* vol, float* tsdf, float* weight, float* color) {
    Vol* v = (Vol*)vol;
    CK(hipMemcpy(tsdf, v->tsdf, v->n * sizeof(float), hipMemcpyDeviceToHost));
    CK(hipMemcpy(weight, v->weight, v->n * sizeof(float), hipMemcpyDeviceToHost));
    CK(hipMemcpy(color, v->color, v->n * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" void tsdf_ref_destroy(void* vol) {
    Vol* v = (Vol*)vol;
    if (!v) return;
    for (float* p : {v->tsdf, v->weight, v->color}) (void)hipFree(p);
    delete v;
}
