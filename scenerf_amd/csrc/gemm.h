// MFMA GEMM family for the ResnetFC pass (gfx950).  See gemm.hip for the kernels.
#pragma once
#include "common.h"

#define GEMM_MAX_SEG 5

// C[M][N] = epilogue( [A1 | A2-segments][M][K] @ W[N][K]^T )
struct GemmNT {
    // operand A, part 1: row-major [M][lda1], first K1 columns; optional relu on load
    const void* A1 = nullptr;
    int lda1 = 0, K1 = 0, relu1 = 0;
    // operand A, part 2: the gathered-feature matrix Z [M][lda2] split in per-scale column segments; a
    // segment is skipped for a 128-row tile when its bit in tile_mask is clear (exact zeros, Q1 in SURVEY §0)
    const void* A2 = nullptr;
    int lda2 = 0, nseg = 0;
    int seg_off[GEMM_MAX_SEG] = {0, 0, 0, 0, 0};
    int seg_len[GEMM_MAX_SEG] = {0, 0, 0, 0, 0};
    const uint8_t* tile_mask = nullptr;
    int skip_bit = -1;  // >= 0: drop the whole row-tile when that mask bit is clear (dZ of one scale)
    // operand W: row-major [N][ldw], columns ordered [K1 | seg0 | seg1 ...]
    const void* W = nullptr;
    int ldw = 0;
    int M = 0, N = 0;
    // epilogue: v = acc + bias[n]; v += res; v = (maskp > 0 ? v : 0); v += res2; store
    const float* bias = nullptr;
    const void* res = nullptr;
    int ldres = 0, res_f32 = 0;
    const void* maskp = nullptr;
    int ldmask = 0;
    const void* res2 = nullptr;
    int ldres2 = 0;
    void* out = nullptr;
    int ldout = 0, out_f32 = 0;
    // scatter epilogue (grid_sampler backward): out is ignored; v * tap_weight is atomically added to
    // gmap[texel][n] for the 4 taps of row m at scale `scatter_scale`
    float* gmap = nullptr;
    long gmap_st = 0, gmap_sc = 1;   // element strides of gmap per texel / per channel: (N, 1) for (H,W,C) [gmap_st == 0 means N], (1, H*W) for (C,H,W)
    const int32_t* tap_texel = nullptr;
    const float* tap_weight = nullptr;
    int scatter_scale = -1;
    // multi-scale scatter (ms_n > 0): ONE launch computes dZ for all scales -- the N axis is the concatenation of the scales' column
    // tiles (ms_t0[s] = first 128-column tile of scale s), W / gmap / strides / column count are taken per scale, skip_bit and
    // scatter_scale become the tile's scale.  All column tiles of a row tile run back to back on one XCD, so the K = 1536 operand
    // rows (dH) are fetched from HBM once instead of once per scale launch.
    int ms_n = 0;
    int ms_t0[GEMM_MAX_SEG + 1] = {0, 0, 0, 0, 0, 0};
    int ms_C[GEMM_MAX_SEG] = {0, 0, 0, 0, 0};
    const void* ms_W[GEMM_MAX_SEG] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    float* ms_gmap[GEMM_MAX_SEG] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // NULL: that scale needs no gradient
    long ms_st[GEMM_MAX_SEG] = {0, 0, 0, 0, 0}, ms_sc[GEMM_MAX_SEG] = {1, 1, 1, 1, 1};
    int force_tile = 0;  // 0 = pick by shape, 1 = 128x128 tile, 2 = 128x512 tile (needs N == 512)
    const char* name = "gemm_nt";
};

// C[N][K] += D[M][N]^T @ act(A[M][K])   (fp32 atomics; contraction over the rows)
struct GemmTN {
    const void* D = nullptr;
    int ldd = 0;
    const void* A = nullptr;
    int lda = 0, relu_a = 0;
    int M = 0, N = 0, K = 0;
    const uint8_t* tile_mask = nullptr;
    int skip_bit = -1;
    float* out = nullptr;
    int ldo = 0;
    float* colsum = nullptr;  // optional [N]: += sum_m D[m][n]  (bias gradient, folded into the k-tile-0 workgroups)
    int allow_tr = 1;         // 0: never the transposing-read kernel (scenerf_cfg.flags & SCENERF_FLAG_NO_WGRAD_TR)
    const char* name = "gemm_tn";
};

int launch_gemm_nt(int precision, const GemmNT& p, hipStream_t s);
int launch_gemm_tn(int precision, const GemmTN& p, hipStream_t s);

// fused.hip: whole ResnetFC trunk (lin_in + lin_z + 3 residual blocks) for bf16 operands in one kernel; writes relu(H_b),
// relu(N_b) (what the backward pass consumes) to a->H / a->Nn.  a->h0pre must hold the split encoding [M][144].
int launch_mlp_fwd_fused(const scenerf_cfg* cfg, const scenerf_mlp_weights* w, const void* Z, const uint8_t* tile_mask, int M,
                         const scenerf_mlp_acts* a, hipStream_t s);
// backward dgrad chain of the three residual blocks in one kernel (bf16): reads dH column block 3, writes dH column blocks 2..0 and
// dN [3][M][512]; sign gates come from the saved activations a->Nn / a->H.
// wide.hip: the same forward on 128-row blocks, one wave per SIMD (activations agree to the last bf16 ulp, not bit for bit: the bias is
// added after the K sum), and the dgrad chain on the same kernel shape (bit-identical to launch_mlp_bwd_fused)
int launch_mlp_fwd_wide(const scenerf_cfg* cfg, const scenerf_mlp_weights* w, const void* Z, const uint8_t* tile_mask, int M,
                        const scenerf_mlp_acts* a, hipStream_t s);
// (d_logits != NULL: dH column block 3 = lin_out's input gradient is made by the kernel itself from d_logits, w_out and H3's sign bits)
int launch_mlp_bwd_wide(const scenerf_cfg* cfg, const scenerf_mlp_weights* w, int M, const scenerf_mlp_acts* a, void* dH, void* dN,
                        const float* d_logits, hipStream_t s);
int launch_mlp_bwd_fused(const scenerf_cfg* cfg, const scenerf_mlp_weights* w, int M, const scenerf_mlp_acts* a, void* dH, void* dN,
                         hipStream_t s);
// wgrad.hip: bf16 weight-gradient GEMM on transposing LDS reads (256 x 256 output tiles, wave-specialised); launch_gemm_tn uses it
// for the shapes it fits
// dfeat.hip: feature-map gradients of one pass (dZ = dH[:, 0:1536] @ Wz scattered through the forward's bilinear taps), bf16
int launch_dfeat_scatter(const scenerf_cfg* cfg, const scenerf_mlp_weights* w, const uint8_t* tile_mask, const int32_t* tap_texel,
                         const float* tap_weight, int M, const void* dH, float* const gmaps[SCENERF_N_SCALES], hipStream_t s);
#define W_SINGLE_MIN_ROWS 32768   // a launch of its own pays the 42-us atomic flush alone
#ifndef W_BATCH_MIN_ROWS
#define W_BATCH_MIN_ROWS 4096     // the batched launch of a backward pass (seven problems, one flush): also the gaussian head's 4,800
                                  // rows -- 65 us instead of eleven per-layer launches of 20-30 us each (r02_d)
#endif
bool wgrad_tr_applicable(const GemmTN& p, int min_rows = W_SINGLE_MIN_ROWS);
int launch_wgrad_tr(const GemmTN& p, hipStream_t s);
int launch_wgrad_tr_batch(const GemmTN* probs, int count, hipStream_t s);   // same (M, N, K) for all: one launch, one atomic flush
