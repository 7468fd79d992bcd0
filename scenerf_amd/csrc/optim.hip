// AdamW over a list of tensors in ONE launch (step-level waste; reference: configure_optimizers, scenerf.py:756-761 --
// torch.optim.AdamW over the module's parameters).  The renderer's two ResnetFC hold 40 parameter tensors / 43.3 MB; torch's fused
// optimizer walks them in two multi-tensor launches at 2.7 TB/s of its 7 x 43.3 MB (104-109 us per step, r02/r03 profiles) plus two
// small copies that make the sliced lin_in gradient contiguous.  This kernel reads gradients where the renderer's gradient sink
// left them (row-strided views included) and streams p, g, m, v through once in 16-byte accesses: HBM-bound, 28 B per element.
//
// Arithmetic = torch.optim.AdamW (amsgrad = False, maximize = False), evaluated in fp32 like torch's fused kernel:
//   p <- p (1 - lr wd);  m <- m + (g - m)(1 - b1);  v <- b2 v + (1 - b2) g g;
//   p <- p - (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// (bias corrections are formed on the host in double, per tensor: a tensor that had no gradient in some step lags behind).
#include "common.h"

#define OPT_MAX_TENSORS 48
#define OPT_CHUNK 4096           // elements per workgroup: 256 threads x 4 float4

struct OptTensor {
    float* p; const float* g; float* m; float* v;
    int n;                        // elements
    int g_cols, g_ld;             // gradient view: rows of g_cols elements at stride g_ld (g_cols == 0: contiguous)
    float step_size, inv_sqrt_bc2;   // lr / (1 - b1^t), 1 / sqrt(1 - b2^t)
};
struct OptArgs {
    OptTensor t[OPT_MAX_TENSORS];
    int first_chunk[OPT_MAX_TENSORS + 1];   // prefix sum of chunks per tensor
    int count;
    float decay, b1, b2, eps;     // decay = 1 - lr * weight_decay
    // capturable form (scenerf_hip_adamw_step_dev): the learning rate and the step count live in device memory, [lr, t, scratch] -- a
    // replayed hipGraph then advances with them; bias corrections are formed here, in fp32, the same for every tensor of the launch.
    // t = steps completed BEFORE this one; the workgroup that finishes last (a counter in the scratch word) stores t + 1 -- every
    // workgroup has read t by then -- so counting the step is neither a launch of its own nor an edge in a captured graph (r04: the
    // one-element add in front of the step was the root of the replayed graph, and the fork behind it cost the chain 14 us)
    float* hyper;
    int bump;                     // this launch counts the step (the last launch of a call)
    float wd;
};

__global__ __launch_bounds__(256) void adamw_kernel(OptArgs a) {
    // which tensor: binary search over the chunk prefix (wave-uniform: blockIdx only)
    int lo = 0, hi = a.count;
    const int b = blockIdx.x;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.first_chunk[mid] <= b) lo = mid; else hi = mid;
    }
    const OptTensor& T = a.t[lo];
    const int e0 = (b - a.first_chunk[lo]) * OPT_CHUNK;
    const float omb1 = 1.f - a.b1, omb2 = 1.f - a.b2;
    float decay = a.decay, step_size = T.step_size, inv_sqrt_bc2 = T.inv_sqrt_bc2;
    if (a.hyper) {
        const float lr = a.hyper[0], t = a.hyper[1] + 1.f;
        decay = 1.f - lr * a.wd;
        step_size = lr / (1.f - powf(a.b1, t));
        inv_sqrt_bc2 = 1.f / sqrtf(1.f - powf(a.b2, t));
    }
#pragma unroll
    for (int it = 0; it < OPT_CHUNK / 1024; ++it) {
        const int e = e0 + it * 1024 + threadIdx.x * 4;
        if (e >= T.n) break;
        float p[4], g[4], m[4], v[4];
        const bool full = e + 4 <= T.n && (T.g_cols == 0);
        if (full) {
            const float4 p4 = *(const float4*)(T.p + e), g4 = *(const float4*)(T.g + e), m4 = *(const float4*)(T.m + e), v4 = *(const float4*)(T.v + e);
            p[0] = p4.x; p[1] = p4.y; p[2] = p4.z; p[3] = p4.w;
            g[0] = g4.x; g[1] = g4.y; g[2] = g4.z; g[3] = g4.w;
            m[0] = m4.x; m[1] = m4.y; m[2] = m4.z; m[3] = m4.w;
            v[0] = v4.x; v[1] = v4.y; v[2] = v4.z; v[3] = v4.w;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = e + q;
                const bool ok = i < T.n;
                const int gi = T.g_cols ? (i / T.g_cols) * T.g_ld + (i % T.g_cols) : i;
                p[q] = ok ? T.p[i] : 0.f; g[q] = ok ? T.g[gi] : 0.f; m[q] = ok ? T.m[i] : 0.f; v[q] = ok ? T.v[i] : 0.f;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            p[q] *= decay;
            m[q] = m[q] + (g[q] - m[q]) * omb1;
            v[q] = a.b2 * v[q] + omb2 * g[q] * g[q];
            const float denom = sqrtf(v[q]) * inv_sqrt_bc2 + a.eps;
            p[q] -= step_size * (m[q] / denom);
        }
        if (full) {
            *(float4*)(T.p + e) = make_float4(p[0], p[1], p[2], p[3]);
            *(float4*)(T.m + e) = make_float4(m[0], m[1], m[2], m[3]);
            *(float4*)(T.v + e) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (e + q < T.n) { T.p[e + q] = p[q]; T.m[e + q] = m[q]; T.v[e + q] = v[q]; }
        }
    }
    if (a.hyper && a.bump) {
        // count the step: every workgroup has read t by the time the last one gets here.  Relaxed device-scope atomics and NO fence: a
        // __threadfence() here is an L2 write-back on this multi-die part, and 2,664 workgroups each flushing the 130 MB this kernel
        // has just written made it 162-198 us instead of 51 (r04).  Nothing needs one: a workgroup's read of t has returned before
        // its atomic is issued (the barrier below), and the stores become visible to the next launch at the kernel boundary.  Two levels
        // of counters (64 groups, then one) keep the returning atomics off a single address.
        __syncthreads();          // (all of this workgroup's reads of t have returned)
        if (threadIdx.x == 0) {
            unsigned* cnt = (unsigned*)(a.hyper + 2);
            const unsigned grp = blockIdx.x & 63u, ngrp = gridDim.x < 64u ? gridDim.x : 64u;
            const unsigned in_grp = (gridDim.x - grp + 63u) >> 6;
            if (__hip_atomic_fetch_add(cnt + 1 + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_grp - 1) {
                __hip_atomic_store(cnt + 1 + grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngrp - 1) {
                    __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(a.hyper + 1, a.hyper[1] + 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}

static int adamw_launch(int count, const scenerf_adamw_tensor* tensors, float* hyper, float lr, float beta1, float beta2, float eps,
                        float weight_decay, scenerf_stream_t stream) {
    SRF_CHECK(tensors && count > 0, "adamw_step: no tensors");
    hipStream_t s = as_stream(stream);
    for (int base = 0; base < count; base += OPT_MAX_TENSORS) {
        OptArgs a;
        a.count = count - base < OPT_MAX_TENSORS ? count - base : OPT_MAX_TENSORS;
        a.decay = 1.f - lr * weight_decay; a.b1 = beta1; a.b2 = beta2; a.eps = eps;
        a.hyper = hyper; a.wd = weight_decay; a.bump = (hyper && base + OPT_MAX_TENSORS >= count) ? 1 : 0;
        int chunks = 0;
        double bytes = 0;
        for (int i = 0; i < a.count; ++i) {
            const scenerf_adamw_tensor& t = tensors[base + i];
            SRF_CHECK(t.p && t.g && t.m && t.v && t.numel > 0 && t.numel < (1ll << 31) && (hyper || t.step >= 1), "adamw_step: bad tensor %d", base + i);
            SRF_CHECK(t.g_cols == 0 || (t.g_ld >= t.g_cols && t.numel % t.g_cols == 0), "adamw_step: bad gradient view of tensor %d", base + i);
            const double st = hyper ? 1.0 : (double)t.step;   // (unused by the kernel when `hyper` is given)
            const double bc1 = 1.0 - pow((double)beta1, st), bc2 = 1.0 - pow((double)beta2, st);
            a.t[i] = {t.p, t.g, t.m, t.v, (int)t.numel, t.g_cols, t.g_ld, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2))};
            a.first_chunk[i] = chunks;
            chunks += (int)((t.numel + OPT_CHUNK - 1) / OPT_CHUNK);
            bytes += 28.0 * (double)t.numel;
        }
        a.first_chunk[a.count] = chunks;
        SrfLaunchScope ps(s, "adamw", 0, bytes);
        adamw_kernel<<<chunks, 256, 0, s>>>(a);
        SRF_LAUNCH_CHECK("adamw_kernel");
    }
    return 0;
}

extern "C" int scenerf_hip_adamw_step(int count, const scenerf_adamw_tensor* tensors, float lr, float beta1, float beta2, float eps,
                                      float weight_decay, scenerf_stream_t stream) {
    return adamw_launch(count, tensors, nullptr, lr, beta1, beta2, eps, weight_decay, stream);
}

extern "C" int scenerf_hip_adamw_step_dev(int count, const scenerf_adamw_tensor* tensors, float* hyper, float beta1, float beta2,
                                          float eps, float weight_decay, scenerf_stream_t stream) {
    SRF_CHECK(hyper, "adamw_step_dev: hyper is NULL");
    return adamw_launch(count, tensors, hyper, 0.f, beta1, beta2, eps, weight_decay, stream);
}
