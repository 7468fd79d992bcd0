/*
 * scenerf_hip.h -- C ABI of libscenerf_hip.so: the MI355X (gfx950) ray-rendering hot path of SceneRF.
 *
 * Drop-in boundary.  The reference has no FFI: its hot path is eager PyTorch inside
 * `SceneRF.render_rays_batch` (reference scenerf/models/scenerf.py:392-471).  What a maintainer binds
 * instead of those torch ops is this library, through ctypes (INTEGRATION.md shows the stub).  Every entry
 * point takes plain device pointers + sizes + a hipStream_t (as void*); nothing here knows about torch.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in `_host`; all work is enqueued on `stream`
 *     and returns immediately (no hidden synchronisation; the only library-owned device state is per-device
 *     and created once under a lock -- see scenerf_hip_prepare; re-entrant, hipGraph-capturable).  `stream`
 *     must belong to the calling thread's current device (hipSetDevice), like the buffers;
 *   - return value 0 = ok, otherwise an error code; `scenerf_hip_last_error()` gives the message.  Bad
 *     arguments are reported, never abort();
 *   - "rows" M = rays * points-per-ray.  Activation buffers are row-major [M][ld];
 *   - precision: 0 = fp32 everywhere (fp32 MFMA v_mfma_f32_32x32x2_f32), 1 = bf16 GEMM operands /
 *     fp32 accumulate, fp32 everything else (BASELINE.json config 2).  "act" buffers are float (0) or
 *     bf16 (1);
 *   - sample order, sphere indices and the sort permutation are int32 on the device (the reference's
 *     int64 tensors are produced by the Python host when a caller asks for them).
 *
 * Each function names the reference code it replaces (paths relative to the reference repo).
 */
#ifndef SCENERF_HIP_H
#define SCENERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCENERF_HIP_ABI_VERSION 10
#define SCENERF_N_SCALES 5          /* feature maps "1_1","1_2","1_4","1_8","1_16" */
#define SCENERF_D_LATENT 2480       /* 80+160+320+640+1280 (resnetfc d_latent, scenerf.py:100-114) */
#define SCENERF_D_HIDDEN 512
#define SCENERF_Z_DENSE_COLS 256    /* columns [0, 256) of the gathered features Z are always defined (zeros where a row tile misses
                                       the scale); the columns of an untouched (tile, scale) pair beyond that are left unwritten */
#define SCENERF_D_XENC 48           /* PE(39) + viewdir(3), zero-padded to a multiple of 16 */
#define SCENERF_TILE_ROWS 128       /* granularity of the scale-activity mask (Q1 sparsity) */
#define SCENERF_MAX_GAUSSIANS 8
#define SCENERF_MAX_SAMPLES 512

typedef void* scenerf_stream_t;     /* hipStream_t */

/* Constants that reach the hot path (reference scenerf.py:23-116; BundleFusion deltas scenerf_bf.py:85-90,606-608). */
typedef struct scenerf_cfg {
    int32_t n_pts_uni;              /* U */
    int32_t n_gaussians;            /* G */
    int32_t n_pts_per_gaussian;     /* P */
    int32_t n_samples;              /* N = U + G*P (U>0) or G*P; U with SCENERF_FLAG_UNIFORM_ONLY */
    int32_t sphere_W, sphere_H;     /* out_img_W/H */
    float max_sample_depth;         /* D */
    float uni_step;                 /* (D-0.2)/U, utils.py:77 */
    float base_std;                 /* `std` ctor arg */
    float som_sigma;
    float gauss_floor;              /* 1.5 (KITTI) / 0.5 (BundleFusion) */
    float kl_std_floor;             /* 1.5, ray_som_kl.py:83 */
    float v_min, v_fov, h_min, h_fov; /* SphericalMapping, spherical_mapping.py:60-67 */
    int32_t map_C[SCENERF_N_SCALES];  /* 80,160,320,640,1280 */
    int32_t map_H[SCENERF_N_SCALES], map_W[SCENERF_N_SCALES]; /* round(H/s), round(W/s): unet2d_sphere.py:139 */
    int32_t div_H[SCENERF_N_SCALES], div_W[SCENERF_N_SCALES]; /* H//s, W//s: scenerf.py:525 */
    int32_t precision;              /* 0 fp32, 1 bf16 operands */
    int32_t map_chw[SCENERF_N_SCALES]; /* layout of the map handed to gather_features / of the gradient buffer handed to mlp_backward:
                                          0: (H,W,C) copy in the activation type (maps_chw_to_hwc), (H,W,C) fp32 gradient accumulator;
                                          1: scale s is NOT converted -- gather_features reads the caller's fp32 (C,H,W) map and the
                                             feature gradient is scattered into an fp32 (C,H,W) buffer (slow per access, meant for the
                                             coarse scales that quirk Q1 keeps out of range for all but <= 1/s^2 of the sphere);
                                          2: the caller's fp32 (H,W,C) map read in place, (H,W,C) fp32 gradient accumulator: no layout
                                             conversion in either direction (the entry for a producer that emits (H,W,C)) */
    /* kernel-path selection (explicit state of the call, never ambient: no environment variable changes what a call computes) */
    int32_t fused_min_rows;         /* bf16: row count M from which the ResnetFC trunk (forward) and the dgrad chain (backward) each run as
                                       ONE fused kernel; below it the per-layer GEMM path runs.  0 = library default (4096); < 0 = never */
    int32_t fwd_kernel;             /* fused forward variant: 0 = LDS-ring pipeline, 64-row blocks (fused.hip); 2 = 128-row blocks, one wave
                                       per SIMD (wide.hip): same rounding points, the bias added after the K sum instead of before it
                                       (last-ulp differences).  (1 was a register-streamed 64-row variant, removed: refused) */
    uint32_t flags;                 /* SCENERF_FLAG_* */
} scenerf_cfg;
#define SCENERF_FUSED_MIN_ROWS_DEFAULT 4096
#define SCENERF_FLAG_NO_FUSED_BWD    1u  /* dgrad chain as six per-layer GEMMs even where the fused chain applies */
#define SCENERF_FLAG_NO_WGRAD_TR     2u  /* weight gradients through gemm_tn only (no transposing-read batched kernel) */
#define SCENERF_FLAG_DFEAT_PER_SCALE 4u  /* feature-gradient GEMM + scatter as one launch per pyramid level (A/B runs) */
#define SCENERF_FLAG_WGRAD_OVERLAP   8u  /* per-layer backward: weight-gradient GEMMs on an internal side stream */
#define SCENERF_FLAG_DFEAT_GEMM     64u  /* bf16 feature-map gradients through the GEMM family's scatter epilogue (gemm.hip) instead of dfeat.hip (A/B runs) */
#define SCENERF_FLAG_WIDE_ANY_M     32u  /* the 128-row kernels also below 192 row blocks (where the 64-row ones are faster): tests, probes */
#define SCENERF_FLAG_UNIFORM_ONLY  128u  /* the reference's uniform-only branch (scenerf.py:647-650 / scenerf_bf.py:662-665: n_pts_uni == 0 and
                                           * n_pts_per_gaussian == 1): the n_pts_uni uniform samples are the ONLY samples that are rendered
                                           * (n_samples == n_pts_uni); the gaussian head is still evaluated for the KL term */
#define SCENERF_FLAG_WIDE_BWD       16u  /* fused dgrad chain on 128-row blocks, one wave per SIMD (wide.hip) instead of fused.hip's 64-row ring;
                                           * the chain then also makes lin_out's input gradient (dH column block 3) from d_logits and H3's sign bits */
#define SCENERF_FLAG_WIDE_BWD_STAGED 256u /* ... but reads dH column block 3 as linout_bwd wrote it, like fused.hip's chain (tests, A/B runs: bit-identical
                                           * to the per-layer dgrad GEMMs) */
#define SCENERF_FLAG_PACK_FORWARD   512u  /* scenerf_hip_mlp_pack in two calls (bf16 with w_stream; otherwise ignored): this one packs what a forward
                                            * pass reads (untransposed operands, their w_stream blocks, biases) ... */
#define SCENERF_FLAG_PACK_REST     1024u  /* ... this one the rest (transposed operands, their w_stream blocks, w_z_t, the `clear` zeroes): a forward
                                            * can be ordered behind the first call alone */

#define SCENERF_FLAG_BWD_CHAIN_ONLY 2048u /* scenerf_hip_mlp_backward in two calls (fused dgrad chain only; without one the first call does everything and
                                            * the second nothing): this one runs lin_out's backward and the dgrad chain ... */
#define SCENERF_FLAG_BWD_GRADS_ONLY 4096u /* ... this one the weight and feature-map gradients that consume the chain's dH / dN: the caller may order
                                            * other work of its own (another stream's kernels) between the two */

/* Packed ResnetFC operands (built by the host from the nn.Linear parameters).
 * T = float (precision 0) or bf16 (precision 1).  reference scenerf/models/resnetfc.py:88-118,133-164 */
typedef struct scenerf_mlp_weights {
    int32_t d_out;                  /* 4 (mlp) or 2 (mlp_gaussian) */
    const float* w_in;              /* [512][48]  lin_in.weight zero-padded 42->48, always fp32 */
    const float* b_in;              /* [512] */
    const void* w_h[4];             /* T: w_h[0] = [512][2480] lin_z.0 (fp32 mode) or [512][144+2480] = [w_in_hi | w_in_hi | w_in_lo | lin_z.0]
                                       (bf16 mode, split-bf16 lin_in fused in); [512][512+2480] = [fc_1.b | lin_z.(b+1)] b=0,1 ; [512][512] = fc_1.2 */
    const float* b_h[4];            /* [512]: lin_z.0.bias (+ lin_in.bias in bf16 mode) ; fc_1.b.bias + lin_z.(b+1).bias ; fc_1.2.bias */
    const void* w_fc0[3];           /* T: [512][512] blocks.b.fc_0.weight */
    const float* b_fc0[3];
    const float* w_out;             /* [d_out][512] fp32 */
    const float* b_out;             /* [d_out] */
    /* backward operands */
    const void* w_fc0_t[3];         /* T: [512][512] = fc_0.weight^T */
    const void* w_fc1_t[3];         /* T: [512][512] = fc_1.weight^T */
    const void* w_z_t[SCENERF_N_SCALES]; /* T: [C_s][1536] = (cat_b lin_z.b.weight[:, slice_s])^T */
    /* bf16 mode only (NULL otherwise, or NULL to disable the fused kernels): the seven forward operands
     * w_h[0], w_fc0[0], w_h[1], w_fc0[1], w_h[2], w_fc0[2], w_h[3], then the six dgrad operands w_fc1_t[2], w_fc0_t[2],
     * w_fc1_t[1], w_fc0_t[1], w_fc1_t[0], w_fc0_t[0], re-tiled for streaming -- per 16 columns of K one
     * contiguous 16 KiB block [512 rows][32 B] holding the exact LDS image (16-byte halves of row r swapped when
     * (r >> 3) & 1).  SCENERF_W_STREAM_BLOCKS blocks in all. */
    const void* w_stream;
    /* optional (scenerf_hip_mlp_pack only): fp32 buffer of clear_floats elements that the pack launch zero-fills on its way -- the
     * gradient sink (scenerf_mlp_grads) of the backward that will follow: no fill launch of its own */
    float* clear;
    int64_t clear_floats;
} scenerf_mlp_weights;
#define SCENERF_W_STREAM_BLOCKS ((3 * SCENERF_D_XENC + SCENERF_D_LATENT) / 16 + 2 * ((SCENERF_D_HIDDEN + SCENERF_D_LATENT) / 16) + 10 * (SCENERF_D_HIDDEN / 16))

/* Raw ResnetFC parameters exactly as the reference's nn.Linear modules hold them (fp32, row-major [out][in]). */
typedef struct scenerf_mlp_params {
    int32_t d_out;
    const float* lin_in_w;          /* [512][42] */
    const float* lin_in_b;          /* [512] */
    const float* lin_out_w;         /* [d_out][512] */
    const float* lin_out_b;         /* [d_out] */
    const float* fc0_w[3];          /* blocks.b.fc_0.weight [512][512] */
    const float* fc0_b[3];
    const float* fc1_w[3];          /* blocks.b.fc_1.weight [512][512] */
    const float* fc1_b[3];
    const float* linz_w[3];         /* lin_z.b.weight [512][2480] */
    const float* linz_b[3];
} scenerf_mlp_params;

#define SCENERF_WIN_LD 256
/* Gradients of the packed operands (fp32, accumulated with atomics: zero them first). */
typedef struct scenerf_mlp_grads {
    float* w_in;                    /* [512][SCENERF_WIN_LD]: columns 0..47 hold the gradient (42 real + pad); the rest of a row is scratch
                                     * -- the batched weight-gradient kernel works in 256-column tiles and lin_in rides along (wgrad.hip) */
    float* b_in;                    /* [512] */
    float* w_fc0[3];                /* [512][512] */
    float* b_fc0[3];
    float* w_fc1[3];                /* [512][512] */
    float* b_fc1[3];
    float* w_z;                     /* [1536][2480]  (rows 512b.. = lin_z.b) */
    float* b_z;                     /* [1536] */
    float* w_out;                   /* [d_out][512] */
    float* b_out;                   /* [d_out] */
    float* w_in_dense;              /* optional [512][42]: written (not accumulated) at the end of every scenerf_hip_mlp_backward with columns 0..41 of
                                     * w_in as they stand -- lin_in.weight's gradient as a dense tensor (no zeroing needed) */
} scenerf_mlp_grads;

/* Saved activations of one ResnetFC evaluation over M rows (caller-allocated).
 * act = T.  H[b] is the residual stream after adding lin_z.b (b<3) / the final one (b=3); N[b] = fc_0 output. */
typedef struct scenerf_mlp_acts {
    void* H[4];                     /* T [M][512]; H and Nn may be NULL for inference when the fused bf16 kernel runs (>= 4096 rows) */
    void* Nn[3];                    /* T [M][512] */
    float* h0pre;                   /* scratch: fp32 mode [M][512] fp32 lin_in output ; bf16 mode the split-bf16 encoding [M + 1][144] bf16 -- ONE ROW OF
                                     * SLACK, zero-filled by the caller: lin_in's weight gradient reads the 144-column rows in 256-column tiles, so the
                                     * last row's tile ends 224 bytes behind [M][144] (what it reads there lands in the scratch columns of
                                     * scenerf_mlp_grads.w_in, which are undefined but must stay finite for a caller that reduces the whole sink) */
    float* logits;                  /* fp32 [M][d_out] */
    uint8_t* sign_bits;             /* bf16 mode, may be NULL: [7][Mpad][64] -- bit n of row m = (saved activation [m][n] > 0) for
                                     * H0, N0, H1, N1, H2, N2, H3; written by the fused forward kernels, gates the fused backward
                                     * chain and (H3's) lin_out's input gradient (Mpad = M rounded up to SCENERF_TILE_ROWS).  NULL disables
                                     * the fused backward. */
    int32_t x3_ready;               /* bf16 mode: h0pre already holds the split encoding (scenerf_hip_encode_points wrote it): the forward
                                     * then neither reads xenc nor launches the split */
} scenerf_mlp_acts;

int scenerf_hip_abi_version(void);
const char* scenerf_hip_last_error(void);
/* Returns and clears the HIP runtime's "last error" of the calling thread.  For callers that catch a FAILED stream capture and go on
 * issuing eagerly: the capture's error otherwise stays behind as the last error and the next entry point of this library -- each checks
 * hipGetLastError() behind its launches -- reports it for a launch that succeeded. */
int scenerf_hip_clear_last_error(void);
/* The id of the stream capture `stream` is recording into, 0 when it is not capturing.  For host-side caches of device buffers: a buffer
 * written inside a capture lives in that graph's memory and only has its contents during a replay, so a cache entry made under capture id
 * X may be reused under X only (scenerf_amd/renderer.py: the converted feature maps of one image, shared by its S source frames). */
int scenerf_hip_stream_capture_id(scenerf_stream_t stream, unsigned long long* id);
/* A stream of the LOWEST priority the device offers (hipDeviceGetStreamPriorityRange: numerically the largest value; PyTorch's own
 * streams stop at the default priority, which is not the lowest on this hardware).  For work that should only take compute units nobody
 * else is waiting for: the trainer's metric-only renders beside the trained ones (scenerf_amd/training.py).  *priority receives the value
 * used, *is_lower whether it is below the default (0).  Destroy with scenerf_hip_stream_destroy (never while work is pending on it). */
int scenerf_hip_stream_create_lowest_priority(scenerf_stream_t* stream, int* priority, int* is_lower);
int scenerf_hip_stream_destroy(scenerf_stream_t stream);

/* One-time setup for the CURRENT device (hipSetDevice) and this configuration: kernel attributes (dynamic LDS sizes), the chunk-descriptor
 * tables of the fused ResnetFC kernels (uploaded asynchronously on `stream`), the zero page.  Every entry point does this lazily on
 * first use, per device and thread-safe; call it explicitly before capturing a hipGraph (the first-use allocations are not capturable).
 * The library keeps no other state: descriptors and attributes are per device ordinal, everything else lives in caller buffers. */
int scenerf_hip_prepare(const scenerf_cfg* cfg, scenerf_stream_t stream);

/* ---- feature-map layout ------------------------------------------------------------------------------- */
/* (C,H,W) fp32 encoder output -> (H,W,C) act (bf16 or fp32): one sample's 2x2 gather becomes 4 contiguous
 * channel runs.  Replaces the CHW strided reads of F.grid_sample, utils.py:239-245. */
int scenerf_hip_maps_chw_to_hwc(const float* chw, void* hwc, int C, int H, int W, int precision, scenerf_stream_t stream);
/* (H,W,C) fp32 gradient accumulator -> (C,H,W) fp32 (grid_sampler_2d_backward's output layout). */
int scenerf_hip_grads_hwc_to_chw(const float* hwc, float* chw, int C, int H, int W, scenerf_stream_t stream);
/* Zero `bytes` (a multiple of 16, dst 16-byte aligned) with `workgroups` workgroups of streaming stores: the fill of the map-gradient
 * accumulators, sized so that it can run beside the forward's latency-bound kernels. */
int scenerf_hip_fill_zero(void* dst, int64_t bytes, int workgroups, scenerf_stream_t stream);

/* ---- ray / sample geometry ----------------------------------------------------------------------------- */
/* utils.py:177-182 (unit dirs), utils.py:112-173 + 75-90 (uniform distances, un-normalised viewdir in the
 * infer frame).  lin_u = torch.linspace(0.2, D, U); noise_u in [0,1) is the reference's rand_like.
 * Sampler noise made on the device WITHOUT a generator launch: noise_u == NULL with rng_state = device uint64 [3] {seed, calls so far,
 * scratch}: ray_setup makes the uniform noise itself (Philox4x32-10 on (element, call), 24-bit uniforms like torch.rand) and
 * scenerf_hip_gaussian_sample_sort, handed the same rng_state, makes the normal noise (Box-Muller), writes it to noise_g for the backward
 * and advances the call counter: one ray_setup + one gaussian_sample_sort per chunk, in this order, on one stream.  Capturable. */
int scenerf_hip_ray_setup(const scenerf_cfg* cfg, const float* pixels /*[R][2]*/, const float* inv_K /*[9]*/,
                          const float* T_s2i /*[16]*/, const float* lin_u /*[U]*/, const float* noise_u /*[R][U] or NULL*/,
                          uint64_t* rng_state /*[3] or NULL*/, int R, float* unit_dir /*[R][3]*/, float* viewdir /*[R][3]*/,
                          float* dist_u /*[R][U]*/, scenerf_stream_t stream);

/* scenerf.py:505-520 up to the gather: points = T @ (dist * unit_dir) (utils.py:158-166), cam_pts_2_pix
 * (utils.py:298-315), SphericalMapping.from_pixels (spherical_mapping.py:80-115, round-half-even),
 * PositionalEncoding (pe.py:32-43).  dist index = ray*dist_ray_stride + (row % pts_per_ray)
 * (stride 0 broadcasts the G anchor distances, scenerf.py:554-572). */
int scenerf_hip_encode_points(const scenerf_cfg* cfg, const float* dist, int dist_ray_stride, int pts_per_ray,
                              const float* unit_dir, const float* viewdir, const float* K /*[9]*/,
                              const float* inv_K /*[9]*/, const float* T_s2i /*[16]*/, int M,
                              float* pts /*[M][3] or NULL*/, int32_t* sphere_idx /*[M][2]*/,
                              float* xenc /*[M][48]; may be NULL when x3 is given*/,
                              void* x3 /*bf16 [M + 1][144] or NULL: the split encoding [hi | lo | hi] of scenerf_mlp_acts.h0pre, written directly
                                         (set scenerf_mlp_acts.x3_ready); the slack row M is zero-filled*/,
                              scenerf_stream_t stream);

/* SphericalMapping.from_pixels (spherical_mapping.py:80-97 -> cam_pts_2_sphere_coords :99-115) for M pixels (u, v): unproject at depth 1
 * with inv_K, angles, `round().long()`.  The rule for the two angles is the one scenerf_hip_encode_points uses for the per-sample index
 * (csrc/sphere_exact.h: torch-CPU's operation sequence with SLEEF's 1.0-ULP acosf / atan2f), so the sphere map the encoder fills and the
 * texels the renderer reads agree.  dist (may be NULL) = the norm of the unprojected point, the reference's third return value. */
int scenerf_hip_pixels_to_sphere(const float* pix /*[M][2]*/, const float* inv_K /*[9]*/, float v_min, float v_fov, float h_min,
                                 float h_fov, int sphere_W, int sphere_H, int64_t M, int64_t* sphere_idx /*[M][2]*/,
                                 float* dist /*[M] or NULL*/, scenerf_stream_t stream);

/* utils.py:232-247 x5 (scenerf.py:522-527): bilinear 2x2 gather of the 5 maps at idx/div*2-1 with zeros
 * padding.  Writes Z rows only for (128-row tile, scale) pairs that have at least one in-range tap and
 * records that in tile_mask (bit s); taps = {texel index or -1, weight} per (row, scale, tap) for backward. */
int scenerf_hip_gather_features(const scenerf_cfg* cfg, const void* const maps_hwc[SCENERF_N_SCALES],
                                const int32_t* sphere_idx, int M, void* Z /*act [Mpad][2480]*/,
                                uint8_t* tile_mask /*[Mpad/128]*/, int32_t* tap_texel /*[M][5][4]*/,
                                float* tap_weight /*[M][5][4]*/, scenerf_stream_t stream);

/* ---- radiance MLP ---------------------------------------------------------------------------------------- */
/* Fill the packed operand buffers of `dst` (caller-allocated, sizes as documented in scenerf_mlp_weights; b_fc0 / w_out /
 * b_out / b_in may simply alias the parameters) from the raw nn.Linear parameters: casts, concatenations along K, the
 * transposed dgrad copies and the split-bf16 lin_in, in two launches.  resnetfc.py:88-118. */
int scenerf_hip_mlp_pack(const scenerf_cfg* cfg, const scenerf_mlp_params* params, const scenerf_mlp_weights* dst,
                         scenerf_stream_t stream);
/* ResnetFC.forward, resnetfc.py:133-164.  Z/xenc as produced above.  Writes acts (kept for backward). */
int scenerf_hip_mlp_forward(const scenerf_cfg* cfg, const scenerf_mlp_weights* w, const void* Z, const float* xenc,
                            const uint8_t* tile_mask, int M, const scenerf_mlp_acts* acts, scenerf_stream_t stream);

/* autograd of ResnetFC.forward w.r.t. parameters and the gathered features; the feature gradient is
 * scattered straight into the (H,W,C) fp32 map-gradient accumulators (grid_sampler_2d_backward).
 * d_logits [M][d_out] fp32.  scratch: dH act [M][2048], dN act [3][M][512] (one per block: the weight-gradient GEMMs
 * run on an internal side stream, forked from / joined into `stream`, while the dgrad chain proceeds). */
int scenerf_hip_mlp_backward(const scenerf_cfg* cfg, const scenerf_mlp_weights* w, const scenerf_mlp_grads* g,
                             const void* Z, const float* xenc, const uint8_t* tile_mask,
                             const int32_t* tap_texel, const float* tap_weight, int M,
                             const scenerf_mlp_acts* acts, const float* d_logits, void* dH, void* dN,
                             float* const gmaps_hwc[SCENERF_N_SCALES] /* may be NULL: skip map grads */,
                             scenerf_stream_t stream);

/* The feature-gradient part of scenerf_hip_mlp_backward on its own: call scenerf_hip_mlp_backward with gmaps_hwc = NULL, start the
 * gradient all-reduce of the (now final) parameter gradients, then scatter the feature gradients from the same dH while the
 * collective is in flight (data-parallel training).  dH as written by scenerf_hip_mlp_backward. */
int scenerf_hip_mlp_feature_grads(const scenerf_cfg* cfg, const scenerf_mlp_weights* w, const uint8_t* tile_mask,
                                  const int32_t* tap_texel, const float* tap_weight, int M, const void* dH,
                                  float* const gmaps_hwc[SCENERF_N_SCALES], scenerf_stream_t stream);

/* ---- probabilistic depth sampler ------------------------------------------------------------------------- */
/* scenerf.py:585-596 (means/stds from the gaussian head), utils.py:186-229 (reparameterised samples,
 * clamp at 0.1), scenerf.py:636-659 (merge with the uniform samples, argsort, gathers).  perm is the
 * stable ascending order (ties by original index).  z = dist * unit_dir.z (depth in the source frame). */
int scenerf_hip_gaussian_sample_sort(const scenerf_cfg* cfg, const float* offsets /*[R][G][2]*/,
                                     const float* anchors /*[G]*/, const float* dist_u /*[R][U]*/,
                                     float* noise_g /*[R][G*P]: read; WRITTEN when rng_state is given*/, uint64_t* rng_state /*[3] or NULL*/,
                                     const float* unit_dir, int R,
                                     float* gmeans /*[R][G]*/, float* gstds /*[R][G]*/, float* dist_sorted /*[R][N]*/,
                                     float* z_sorted /*[R][N]*/, int32_t* perm /*[R][N]*/, scenerf_stream_t stream);

/* ---- alpha compositing ------------------------------------------------------------------------------------ */
/* scenerf.py:535-536 (sigmoid / softplus(x-1) heads) + render_depth_and_color scenerf.py:704-748. */
int scenerf_hip_composite_forward(const float* logits /*[R*N][4]*/, const float* dist_sorted, const float* z_sorted,
                                  int R, int N, float* densities, float* alphas, float* weights /*[R][N]*/,
                                  float* depth /*[R]*/, float* color /*[R][3]*/, float* closest /*[R]*/,
                                  float* weights_at_depth /*[R]*/, int32_t* closest_idx /*[R]*/,
                                  scenerf_stream_t stream);
/* its autograd: upstream grads of depth/color (required) and weights/alphas/densities/depth_volumes (NULL = 0). */
int scenerf_hip_composite_backward(const float* logits, const float* dist_sorted, const float* z_sorted, int R, int N,
                                   const float* g_depth, const float* g_color, const float* g_weights,
                                   const float* g_alphas, const float* g_densities, const float* g_zvol,
                                   float* d_logits /*[R*N][4]*/, float* d_dist /*[R][N]*/, float* d_z /*[R][N]*/,
                                   scenerf_stream_t stream);

/* ---- RaySOM KL ---------------------------------------------------------------------------------------------- */
/* RaySOM.forward + kl_gauss, ray_som_kl.py:10-87.  kl_saved [R][G][3] = {som mean, clamped som std, mask}.
 * bmu_out (nullable, [R][N] bytes): the best-matching unit chosen per sample (ray_som_kl.py:52, the argmax) -- a discrete choice
 * exported so that a parity test can compare it on its own and evaluate its checker at the same choice. */
int scenerf_hip_raysom_forward(const scenerf_cfg* cfg, const float* gmeans, const float* gstds, const float* dist_sorted,
                               const float* alphas, int R, float* loss_kl /*[R]*/, float* som_means /*[R][G]*/,
                               float* som_vars /*[R][G]*/, float* kl_saved /*[R][G][3]*/, uint8_t* bmu_out,
                               scenerf_stream_t stream);

/* The per-ray tail of a chunk in ONE launch each way (what RenderChunk calls): compositing + RaySOM forward (scenerf.py:704-748,
 * ray_som_kl.py:10-87 -- the sorted distances and fresh alphas stay in the wave's registers for the SOM update), and their autograd
 * together with the sampler's (compositing backward + reparameterisation / relu / kl_gauss backward: the gradients w.r.t. the sorted
 * distances and depths never leave the wave; d_dist / d_z are written only if non-NULL).  Results are bit-identical to the stage
 * entries above and below, which remain for per-stage use.  Algorithmic HBM bytes per ray: 32 N + 24 forward, 44 N + 40 backward.
 * Forward, optional parts (a no_grad render that asked for a subset of the outputs): loss_kl == NULL (then som_means, som_vars, kl_saved
 * and bmu_out must be NULL too; gmeans / gstds are not read) = compositing only, no RaySOM pass; densities / alphas / weights are each
 * written only if non-NULL (alphas is required with the RaySOM half): a depth + colour render moves 20 N + 24 bytes per ray. */
int scenerf_hip_ray_tail_forward(const scenerf_cfg* cfg, const float* logits, const float* dist_sorted, const float* z_sorted,
                                 const float* gmeans, const float* gstds, int R, float* densities, float* alphas, float* weights,
                                 float* depth, float* color, float* closest, float* weights_at_depth, int32_t* closest_idx,
                                 float* loss_kl, float* som_means, float* som_vars, float* kl_saved, uint8_t* bmu_out,
                                 scenerf_stream_t stream);
int scenerf_hip_ray_tail_backward(const scenerf_cfg* cfg, const float* logits, const float* dist_sorted, const float* z_sorted, int R,
                                  const float* g_depth, const float* g_color, const float* g_weights, const float* g_alphas,
                                  const float* g_densities, const float* g_zvol, const float* offsets, const float* anchors,
                                  const float* noise_g, const float* unit_dir, const float* gmeans, const float* gstds,
                                  const int32_t* perm, const float* kl_saved, const float* g_loss_kl, const float* g_gmeans,
                                  const float* g_gstds, float* d_logits /*[R*N][4]*/, float* d_offsets /*[R][G][2]*/,
                                  float* d_dist /*[R][N] or NULL*/, float* d_z /*[R][N] or NULL*/,
                                  const float* g_weights_at_depth /*[R] or NULL*/, const float* g_closest /*[R] or NULL*/,
                                  const int32_t* closest_idx /*[R]: the forward's argmin, needed with either of the two*/,
                                  scenerf_stream_t stream);

/* autograd of the sampler + KL w.r.t. the gaussian-head outputs: reparameterisation (utils.py:213, not
 * through the 0.1 clamp), z = dist*unit.z, relu of scenerf.py:591-594, kl_gauss(m1,s1).  Upstream NULL = 0. */
int scenerf_hip_sampler_backward(const scenerf_cfg* cfg, const float* offsets, const float* anchors, const float* noise_g,
                                 const float* unit_dir, const float* gmeans, const float* gstds, const int32_t* perm,
                                 const float* d_dist, const float* d_z, const float* kl_saved, const float* g_loss_kl,
                                 const float* g_gmeans, const float* g_gstds, int R, float* d_offsets /*[R][G][2]*/,
                                 scenerf_stream_t stream);

/* ---- loss-side gathers fused with the renderer's per-ray outputs ------------------------------------------ */
/* scenerf.py:302-307 (colour L1 against sample_pix_features of the source image, utils.py:250-266) and compute_reprojection_loss,
 * scenerf.py:349-386 (unproject at the rendered depth, transform, project utils.py:298-315, sample the target image, min(L1, identity
 * L1 + noise), mean over the rays whose target point has z > 0), one thread per ray.  Everything is device memory, fp32: cam_K /
 * inv_K row-major 3x3, T_source2target row-major 4x4.  noise [R] may be NULL.  Outputs: loss_color [R][3],
 * loss_reprojection [1]; ray_term / valid / dterm_ddepth [R], col_src [R][3], acc2 [2] are kept for the backward. */
int scenerf_hip_loss_side_forward(const float* pix, const float* color, const float* depth, const float* img_source,
                                  const float* img_target, const float* noise, const float* cam_K, const float* inv_K,
                                  const float* T_source2target, int R, int H, int W, float* loss_color, float* ray_term, float* valid,
                                  float* dterm_ddepth, float* col_src, float* acc2, float* loss_reprojection, scenerf_stream_t stream);
/* its autograd w.r.t. the rendered colour and depth (the images carry no gradient in the reference).  Upstream NULL = 0. */
int scenerf_hip_loss_side_backward(const float* color, const float* col_src, const float* valid, const float* dterm_ddepth,
                                   const float* acc2, const float* g_loss_color, const float* g_loss_reprojection, int R,
                                   float* g_color, float* g_depth, scenerf_stream_t stream);

/* The whole loss of ONE source frame in one launch each way (reference scenerf.py:203-238, the weights of forward(), around
 * process_single_source :243-320; BundleFusion weights scenerf_bf.py:215,238):
 *   total = w_rep * loss_reprojection + w_col * mean(loss_color) + mean(loss_kl) + w_d2c * mean_r min_k |gaussian_means[r][k] - depth[r]|
 * with loss_color / loss_reprojection as in scenerf_hip_loss_side_forward and the rendered depth detached in the last term (:287-290).
 * loss_kl [R], gmeans / gstds / som_vars [R][G] are the renderer's outputs (gstds, som_vars nullable: they only feed the two logged means).
 * noise [R] nullable: noise[r] * noise_scale is added to the identity term (the reference draws randn * 1e-5); with noise == NULL and
 * rng_state = device uint64 {seed, calls so far} the N(0,1) values are made in the kernel (Philox4x32-10 on (ray, call), Box-Muller) and the
 * call counter is advanced by the launch itself (capturable: a replayed hipGraph draws fresh noise).  out8 (device, fp32 [8]) =
 * {total, loss_reprojection, mean loss_color, mean loss_kl, mean dist-to-closest-gaussian, mean som_vars of the closest gaussian, mean
 * gaussian_stds of the closest gaussian, number of valid rays}.  Kept for the backward: valid, dterm_ddepth [R], col_src [R][3],
 * closest [R] (int32); partial: scratch, fp32 [8 * ceil(R / 64)].  Sums are taken in a fixed order (no atomics). */
int scenerf_hip_source_loss_forward(const float* pix, const float* color, const float* depth, const float* loss_kl, const float* gmeans,
                                    const float* gstds, const float* som_vars, int G, const float* img_source, const float* img_target,
                                    const float* noise, uint64_t* rng_state /* device [2] or NULL */, float noise_scale, const float* cam_K,
                                    const float* inv_K, const float* T_source2target, int R, int H, int W, float w_rep, float w_col, float w_d2c, float* valid,
                                    float* dterm_ddepth, float* col_src, int32_t* closest, float* partial, float* out8,
                                    float* total /* [1]: out8[0] once more, in a buffer of its own */, scenerf_stream_t stream);
/* its autograd w.r.t. colour [R][3], depth [R], loss_kl [R] and gaussian_means [R][G]; g_total: device scalar, NULL = 1. */
int scenerf_hip_source_loss_backward(const float* color, const float* col_src, const float* valid, const float* dterm_ddepth,
                                     const float* gmeans, const float* depth, const int32_t* closest, const float* out8, const float* g_total,
                                     int R, int G, float w_rep, float w_col, float w_d2c, float* g_color, float* g_depth, float* g_loss_kl,
                                     float* g_gmeans, scenerf_stream_t stream);

/* The depth metrics of one evaluation (reference scenerf/loss/depth_metrics.py:3-24, called per source frame from scenerf.py:322-346 /
 * scenerf_bf.py:340-366): pred clamped to [min_depth, max_depth], then abs_rel, sq_rel, rmse, rmse_log and the three threshold
 * accuracies (max(gt/pred, pred/gt) < 1.25^k) as means over the n entries -- over the entries with mask[i] != 0 when mask is given (bytes;
 * == the reference's gt[mask], pred[mask]; an empty selection gives zeros).  out8 (device fp32 [8]) = {abs_rel, sq_rel, rmse, rmse_log,
 * a1, a2, a3, number of entries used}.  One launch of one block, sums in double in a fixed order (no atomics): replaces ~35 elementwise /
 * reduction launches per source frame of the trainer's step. */
int scenerf_hip_depth_errors(const float* gt, const float* pred, const unsigned char* mask /* nullable */, int64_t n, float min_depth,
                             float max_depth, float* out8, scenerf_stream_t stream);

/* ---- ResnetFC of any shape, forward only (fp32) ---------------------------------------------------------------- */
/* resnetfc.py:67-164 is generic in n_blocks / d_hidden; SceneRF instantiates 3 x 512 (scenerf.py:100-114) and the fast kernels above are
 * built for that.  BASELINE.json configs[0] names a 1-block, 128-wide net: this entry evaluates ANY ResnetFC(d_in=42, d_latent=2480) on the
 * outputs of scenerf_hip_encode_points / _gather_features with one fp32-MFMA GEMM per nn.Linear, in the reference's order of additions.
 * Parameters in the reference's layout (nn.Linear.weight = [out][in], row-major fp32) except: w_in zero-padded 42 -> 48 input columns,
 * w_out / b_out zero-padded to d_out_pad rows (a multiple of 8).  h_a, h_b, n_buf: [M][d_hidden] fp32 scratch; logits [M][d_out_pad]. */
#define SCENERF_RESNETFC_MAX_BLOCKS 8
typedef struct {
    int32_t n_blocks, d_hidden, d_out_pad;
    const float* w_in;  /* [d_hidden][48] */
    const float* b_in;
    const float* w_z[SCENERF_RESNETFC_MAX_BLOCKS];    /* lin_z.b.weight [d_hidden][2480] */
    const float* b_z[SCENERF_RESNETFC_MAX_BLOCKS];
    const float* w_fc0[SCENERF_RESNETFC_MAX_BLOCKS];  /* blocks.b.fc_0.weight [d_hidden][d_hidden] */
    const float* b_fc0[SCENERF_RESNETFC_MAX_BLOCKS];
    const float* w_fc1[SCENERF_RESNETFC_MAX_BLOCKS];
    const float* b_fc1[SCENERF_RESNETFC_MAX_BLOCKS];
    const float* w_out; /* [d_out_pad][d_hidden] */
    const float* b_out; /* [d_out_pad] */
} scenerf_resnetfc;
int scenerf_hip_resnetfc_forward(const scenerf_cfg* cfg, const scenerf_resnetfc* net, const float* xenc /*[M][48]*/,
                                 const float* Z /*[Mpad][2480] fp32, as scenerf_hip_gather_features wrote it*/, const uint8_t* tile_mask,
                                 int M, float* h_a, float* h_b, float* n_buf, float* logits, scenerf_stream_t stream);

/* ---- ResnetFC of any shape, trainable (fp32, per-layer GEMMs; round 6) -------------------------------------------------------
 * resnetfc.py:133-164 with its autograd, for every block count / hidden width the class allows: the forward below additionally SAVES what a
 * backward pass reads -- per block b the pre-activations hz[b] = h + lin_z.b(z) (the input of fc_0) and n[b] = fc_0(relu(hz[b])) (the input
 * of fc_1), and h_fin (the input of lin_out), each [M][d_hidden] fp32 -- and the backward runs one fp32-MFMA GEMM per gradient:
 *   lin_out:  dW_out += dlog^T relu(h_fin), db_out += colsum(dlog), dh = (dlog W_out) * [h_fin > 0]
 *   block b (last to first):  dW1 += dh^T relu(n[b]), db1 += colsum(dh);  dn = (dh W1) * [n[b] > 0];  dW0 += dn^T relu(hz[b]), db0 += colsum(dn);
 *                             dhz[b] = dh + (dn W0) * [hz[b] > 0];  dh = dhz[b]
 *   lin_z:    dWz[b] += dhz[b]^T z (per pyramid level, row tiles without that level skipped), lin_in: dW_in += dhz[0]^T x, db_in += colsum(dhz[0])
 *   maps:     dz = [dhz[0] | .. | dhz[nb-1]] Wz, scattered through the forward's bilinear taps (grid_sample backward)
 * (db_z[b] equals db_in for b = 0 and db1 of block b - 1 otherwise -- the same column sums: the caller copies them.)
 * `net` additionally needs the transposed operands of the input-gradient GEMMs.  Gradients ACCUMULATE into `grads` (the caller zeroes).
 * dlog16: d_logits zero-padded to 16 columns; dhz: [M][n_blocks * d_hidden] (output, also scratch); dh, dn: [M][d_hidden] scratch. */
typedef struct {
    const float* w_fc0_t[SCENERF_RESNETFC_MAX_BLOCKS];  /* fc_0.weight^T: [d_hidden(in)][d_hidden(out)] -> as GEMM operand [n = in][k = out] */
    const float* w_fc1_t[SCENERF_RESNETFC_MAX_BLOCKS];
    const float* w_out_t;                               /* [d_hidden][16]: lin_out.weight^T, output columns zero-padded to 16 */
    const float* w_z_t[SCENERF_N_SCALES];               /* per pyramid level: [map_C[s]][n_blocks * d_hidden] = columns of [Wz_0; ..; Wz_{nb-1}]^T */
} scenerf_resnetfc_t;
typedef struct {
    float* hz[SCENERF_RESNETFC_MAX_BLOCKS];
    float* n[SCENERF_RESNETFC_MAX_BLOCKS];
    float* h_fin;
} scenerf_resnetfc_acts;
typedef struct {
    float* w_in;  /* [d_hidden][48] */
    float* b_in;
    float* w_z;   /* [n_blocks * d_hidden][2480]: lin_z.b.weight's gradient = rows [b * d_hidden, (b + 1) * d_hidden) */
    float* w_fc0[SCENERF_RESNETFC_MAX_BLOCKS];
    float* b_fc0[SCENERF_RESNETFC_MAX_BLOCKS];
    float* w_fc1[SCENERF_RESNETFC_MAX_BLOCKS];
    float* b_fc1[SCENERF_RESNETFC_MAX_BLOCKS];
    float* w_out; /* [16][d_hidden] (rows >= d_out: padding) */
    float* b_out; /* [16] */
} scenerf_resnetfc_grads;
int scenerf_hip_resnetfc_forward_train(const scenerf_cfg* cfg, const scenerf_resnetfc* net, const float* xenc, const float* Z,
                                       const uint8_t* tile_mask, int M, const scenerf_resnetfc_acts* acts, float* h_scratch /*[M][d_hidden]*/,
                                       float* logits /*[M][d_out_pad]*/, scenerf_stream_t stream);
int scenerf_hip_resnetfc_backward(const scenerf_cfg* cfg, const scenerf_resnetfc* net, const scenerf_resnetfc_t* net_t,
                                  const scenerf_resnetfc_grads* grads, const float* xenc, const float* Z, const uint8_t* tile_mask,
                                  const int32_t* tap_texel, const float* tap_weight, int M, const scenerf_resnetfc_acts* acts,
                                  const float* dlog16 /*[M][16]*/, float* dhz, float* dh, float* dn,
                                  float* const gmaps_hwc[SCENERF_N_SCALES] /*NULL: no map gradients*/, scenerf_stream_t stream);

/* ---- generic building blocks exported for unit tests ------------------------------------------------------- */
/* C[M][N] = relu?(A[M][K]) @ W[N][K]^T (+bias); act operands per `precision`, fp32 output.
 * tile: 0 = chosen by shape, 1 = 128x128 workgroup tile, 2 = 128x512 (N must be 512). */
int scenerf_hip_test_gemm_nt(int precision, const void* A, const void* W, const float* bias, int M, int N, int K,
                             int relu_a, int tile, float* C, scenerf_stream_t stream);
/* C[N][K] += D[M][N]^T @ relu?(A[M][K]); colsum[N] += column sums of D (may be NULL); fp32 atomics. */
int scenerf_hip_test_gemm_tn(int precision, const void* D, const void* A, int M, int N, int K, int relu_a,
                             float* C, float* colsum, scenerf_stream_t stream);

/* csrc/sphere_exact.h's two routines on their own: acos_a[i] = acos(a[i]) (may be NULL), atan2_ab[i] = atan2(a[i], b[i]) (may be NULL). */
int scenerf_hip_test_acos_atan2(const float* a, const float* b, int64_t n, float* acos_a, float* atan2_ab, scenerf_stream_t stream);

/* Host-only (no GPU needed): the chunk-descriptor tables the fused ResnetFC kernels walk -- kind 0: fused.hip (33 sets: one per
 * scale mask for the forward, set 32 = backward chain), kind 2: wide.hip (33 sets: 32 forward masks + the backward chain; kind 1 is refused).  A set is SCENERF_CHUNK_TABLE_STRIDE ints:
 * entry 0 = number of descriptors, then the descriptors, zero-padded.  Returns the number of ints written (<= cap) or < 0. */
#define SCENERF_CHUNK_TABLE_STRIDE 704
int scenerf_hip_test_chunk_table(const scenerf_cfg* cfg, int kind, int32_t* out, int cap);

/* Two tuning knobs with measured defaults, exported for same-box A/B runs (< 0 leaves a value as it is; results do not depend on either):
 *  warm_wide (default 1): L2 warm-up of the 128-row fused ResnetFC kernels (csrc/wide.hip) -- the blocks of a launch's first dispatch round,
 *    one per CU, all starting together on an L2 that does not hold the launch's weight stream, each touch a slice of that stream before they
 *    start, so that the XCD's L2 holds all of it after one memory round trip instead of missing chunk by chunk in lockstep;
 *  dfeat_delay_us (default 12): scenerf_hip_mlp_backward with SCENERF_FLAG_WGRAD_OVERLAP holds the feature-gradient launch back by this many
 *    microseconds behind the batched weight-gradient launch it runs beside -- that launch is ONE round of workgroups that each need most of a
 *    CU's LDS, and if the 1,200 smaller feature-gradient workgroups reach the CUs first its last workgroups are placed only as those
 *    drain (measured: 1,040 us instead of 675). */
int scenerf_hip_test_set_tuning(int warm_wide, int dfeat_delay_us);

/* ---- in-library kernel timing (bench.py's roofline leg) ----------------------------------------------------- */
/* While enabled every kernel launch is bracketed by hipEvents on its own stream. */
int scenerf_hip_profile_enable(int on);
/* After a stream sync: copy up to `cap` records {name, launches, total_ms, flops, bytes}. Returns count. */
typedef struct scenerf_prof_rec {
    char name[48];
    int32_t launches;
    float total_ms;
    double flops;                   /* algorithmic flops issued by these launches (MFMA kernels) */
    double bytes;                   /* algorithmic bytes moved (HBM-bound kernels) */
} scenerf_prof_rec;
int scenerf_hip_profile_collect(scenerf_prof_rec* out_host, int cap);

/* ---- TSDF fusion (off the render path) ------------------------------------------------- */
/* One RGB-D frame into a voxel volume: the reference's pycuda kernel (scenerf/data/utils/fusion.py:72-145, semantics 0: truncated,
 * normalised distance, running weighted average of distance and colour, fp32) or the CPU path next to it (fusion.py:236-325,
 * semantics 1: metric distance, a voxel keeps the observation of smallest magnitude and its colour, float64 projection with
 * round-half-even; needs cam_pose_inv = inverse(cam_pose) in float64, which the reference computes with numpy).  Volumes:
 * device fp32 [X][Y][Z] C-order, tsdf initialised to 255, weight and colour to 0 (fusion.py:52-56).  color_im: device fp32
 * [im_h][im_w] folded as floor(b*65536 + g*256 + r) (fusion.py:219-220); depth_im: device fp32 [im_h][im_w], 0 = invalid. */
int scenerf_hip_tsdf_integrate(float* tsdf_vol, float* weight_vol, float* color_vol, const int32_t vol_dim[3],
                               const float vol_origin[3], double voxel_size, const float cam_intr[9], const float cam_pose[16],
                               const double cam_pose_inv[16], const float* color_im, const float* depth_im, int im_h, int im_w,
                               float trunc_margin, float obs_weight, int semantics, scenerf_stream_t stream);

/* ---- optimizer step of the renderer's parameters --------------------------------------------------------------------
 * torch.optim.AdamW (amsgrad = False) over `count` tensors in one launch (reference: configure_optimizers, scenerf.py:756-761).
 * p, m (exp_avg), v (exp_avg_sq): contiguous fp32 [numel]; g: fp32, contiguous (g_cols == 0) or rows of g_cols elements at stride
 * g_ld (a sliced view of the gradient sink); step >= 1 = this tensor's step count INCLUDING this step.  `tensors` is a HOST array. */
typedef struct scenerf_adamw_tensor {
    float* p;
    const float* g;
    float* m;
    float* v;
    int64_t numel;
    int32_t g_cols, g_ld;
    int64_t step;
} scenerf_adamw_tensor;
int scenerf_hip_adamw_step(int count, const scenerf_adamw_tensor* tensors, float lr, float beta1, float beta2, float eps,
                           float weight_decay, scenerf_stream_t stream);
/* The capturable form (torch.optim.AdamW(capturable=True)): the learning rate and the step count are read from device memory,
 * hyper = [lr, t, scratch[SCENERF_ADAMW_SCRATCH]] (fp32; scratch zero-initialised by the caller, owned by the kernel): t = the steps completed BEFORE this
 * call, the same for every tensor; the update uses t + 1 and the launch stores t + 1 back when its last workgroup retires, so a
 * hipGraph that holds this launch counts on with every replay and no separate increment is launched.  Bias corrections are formed on
 * the device in fp32.  tensors[i].step is ignored. */
#define SCENERF_ADAMW_SCRATCH 65
int scenerf_hip_adamw_step_dev(int count, const scenerf_adamw_tensor* tensors, float* hyper, float beta1, float beta2, float eps,
                               float weight_decay, scenerf_stream_t stream);

/* ---- image -> sphere resampling of the encoder levels --------------------------------------------------------
 * Replaces DecoderSphere.get_sphere_feature (reference scenerf/models/unet2d_sphere.py:138-165; six calls per image).
 * map_build  = :140-147, the scatter of `pix // scale` into the (out_w x out_h) sphere grid at round(pix_sphere / scale)
 *              (clamped); duplicate cells: the LAST pixel wins (the reference's single-thread CPU result; undefined on its GPU
 *              path).  pix: device fp32 [n_pix][2] (x, y); pix_sphere: device int64 [n_pix][2]; winner: device scratch
 *              int32 [out_h*out_w]; src: device int32 [out_h][out_w] = (sy << 16 | sx) or -1 for an empty cell.  out_w / out_h
 *              are the level's dims, round(out_img_W / scale) and round(out_img_H / scale) (Python round) -- the caller's job.
 * forward    = :149-165 (normalise + F.grid_sample bilinear/zeros/align_corners=False + permute): x device fp32
 *              [planes][H][W] (planes = B*C), out device fp32 [planes][out_h][out_w], written in full (empty cells = 0).
 * backward   = the adjoint w.r.t. x in gather form: row_ptr device int32 [(H+1)*(W+1)+1], cells device int32 [n_mapped] = the
 *              mapped cells grouped by source pixel sy*(W+1)+sx (sy <= H, sx <= W), ascending inside a group; dx device fp32 [planes][H][W], written in full. */
int scenerf_hip_sphere_map_build(const float* pix, const int64_t* pix_sphere, int64_t n_pix, int scale, int out_w, int out_h,
                                 int32_t* winner, int32_t* src, scenerf_stream_t stream);
int scenerf_hip_sphere_resample_forward(const float* x, int64_t planes, int H, int W, const int32_t* src, int out_w, int out_h,
                                        float* out, scenerf_stream_t stream);
int scenerf_hip_sphere_resample_backward(const float* dout, int64_t planes, int H, int W, const int32_t* row_ptr, const int32_t* cells,
                                         int out_w, int out_h, float* dx, scenerf_stream_t stream);
/* The same two with a channels-last sphere side: out / dout are fp32 [B][out_h][out_w][C] (planes = B*C, a multiple of C) -- the layout
 * the renderer reads in place (scenerf_cfg.map_chw == 2), so that no layout conversion sits between the decoder and gather_features. */
int scenerf_hip_sphere_resample_forward_nhwc(const float* x, int64_t planes, int C, int H, int W, const int32_t* src, int out_w, int out_h,
                                             float* out, scenerf_stream_t stream);
int scenerf_hip_sphere_resample_backward_nhwc(const float* dout, int64_t planes, int C, int H, int W, const int32_t* row_ptr,
                                              const int32_t* cells, int out_w, int out_h, float* dx, scenerf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SCENERF_HIP_H */
