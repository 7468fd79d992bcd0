// Shared by the fused ResnetFC kernels (fused.hip: 64-row LDS-ring pipeline; wide.hip: 128-row blocks).
#pragma once
#include "gemm.h"
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_f;
typedef __attribute__((ext_vector_type(16))) float f32x16_f;

#define F_BM 64
#define F_BK 16                           // K elements per chunk (one MFMA k-step)
#define F_AROW 1024                       // A-buffer row: 512 bf16
#define F_ABUF (F_BM * F_AROW)            // 65536
#define F_A2STG (F_BM * F_BK * 2)         // streamed-A stage: 64 rows x 32 B

struct FusedLayer {
    const float* bias;   // forward: [512]
    void* save;          // [M][save_ld] bf16: forward relu(output) (H_b or N_b); backward dN_b or the dH_b column block
    uint8_t* sign;       // [Mpad][64] sign bits of the saved activation: forward writes its layer's, backward reads the gate of its output
    int kind;            // forward 1: n = acc (fc_0), else h += acc ; backward 1: dN = acc*mask, else dh += acc*mask
    int save_ld;         // row stride of `save` in elements
};
struct FusedArgs {
    FusedLayer layer[7];
    const void* Wst;     // w_stream: 16 KiB blocks, see scenerf_hip.h
    const void* X3;      // forward: [M][144] bf16 split encoding
    const void* Z;       // forward: [Mpad][2480] bf16
    const void* dH3;     // backward: incoming gradient rows, [M][dH_ld] bf16 (column block 3 of the dH scratch)
    int dH_ld;
    const float* dlog;   // backward, wide.hip MODE 1: d_logits [M][d_out] fp32 -- dH3 is made from it in the kernel (layer[6] = where it is saved
                         // and H3's sign bits) instead of read

    const uint8_t* tile_mask;
    const int* desc;     // [32][F_MAXCH] per tile mask: header {number of chunks}, chunk descriptors, zero padding
    int M;
    // forward: lin_out fused behind the last layer (resnetfc.py:162-163): logits[m][j] = relu(H3[m]) . w_out[j] + b_out[j]
    const float* w_out;  // [d_out][512] fp32
    const float* b_out;
    float* logits;       // [M][d_out]
    int d_out;
    int warm;            // > 0: the blocks of the first dispatch round touch the launch's weight blocks (srf L2 warm-up, wide.hip)
};

int srf_warm_wide();     // scenerf_hip_test_set_tuning(): 0 = off, 1 = on (default)

// One 32-bit descriptor per 16-wide K chunk (host-built per tile mask, read with scalar loads one step ahead):
//   [0:9] w_stream block   [10:17] A column / 16   [18:19] src (0 = resident A buffer, 1 = X3, 2 = Z)   [20:22] layer
//   [23] last chunk of its layer   [24] first chunk of its layer   [25:27] ring stage (chunk index mod 5)
typedef const __attribute__((address_space(4))) int* desc_ptr;   // constant address space: descriptor reads are s_load
#define FD_Z(d) ((d) & 1023)
#define FD_Y(d) ((((d) >> 10) & 255) * F_BK)
#define FD_SRC(d) (((d) >> 18) & 3)
#define FD_LAYER(d) (((d) >> 20) & 7)
#define FD_END(d) (((d) >> 23) & 1)
#define FD_BEGIN(d) (((d) >> 24) & 1)
#define FD_STAGE(d) (((d) >> 25) & 7)
#define F_MAXCH SCENERF_CHUNK_TABLE_STRIDE   // 704:     // 666 chunks with all five scales + header + zero padding (the pipeline reads a few entries past the end)

// Per-device cache of a host-built descriptor table (one per kernel family): built once per (device, segment layout) and uploaded with
// a blocking copy before its pointer is handed out -- the table is shared by every stream of the device.  A first use inside a
// hipGraph capture is not capturable -- call scenerf_hip_prepare before capturing.
#define SRF_DESC_LAYOUTS 4        // segment layouts (map channel counts) kept per device; the oldest is replaced beyond that
struct SrfDescCache {
    std::mutex mu;
    struct Slot {
        int seg_len[SRF_DESC_LAYOUTS][5];
        int* d_desc[SRF_DESC_LAYOUTS] = {nullptr, nullptr, nullptr, nullptr};
        int next = 0;
    } slot[SRF_MAX_DEVICES];
};
int srf_desc_cache_get(SrfDescCache& C, const scenerf_cfg* cfg, hipStream_t s, int (*build)(const scenerf_cfg*, std::vector<int>&),
                       const int** desc);
// one-time per-device setup of each kernel family (kernel attributes, descriptor tables, the zero page)
int fused_prepare(const scenerf_cfg* cfg, hipStream_t s);
int wide_prepare(const scenerf_cfg* cfg, hipStream_t s);
int wgrad_prepare();
int gemm_prepare();

// host-only builders of the chunk-descriptor tables (also reachable through scenerf_hip_test_chunk_table for the CPU tests)
int fused_table_build(const scenerf_cfg* cfg, std::vector<int>& tab);    // fused.hip: 33 sets of F_MAXCH ints
int wide_table_build(const scenerf_cfg* cfg, std::vector<int>& tab);     // wide.hip: 33 sets of F_MAXCH ints (32 forward masks + the backward chain)
