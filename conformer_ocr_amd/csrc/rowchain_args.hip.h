// Arguments of the row-chain kernels (rowchain.hip.h) and the entry points of their two translation units (rowchain_d256.hip,
// rowchain_d512.hip: the template instantiations are compiled beside cocr_api.hip, in parallel).
#pragma once
#include <hip/hip_runtime.h>

#include "common.hip.h"

enum { ST_ROWLN = 0, ST_FFN = 1, ST_GLU = 2, ST_QKV = 3, ST_FRONT = 4 };

struct ChainStage {
    int kind;
    const bf16_t *W;       // fragment-major copies.  ROWLN / GLU / QKV: (N, D); FFN: W1 (FF, D)
    const bf16_t *W2;      // FFN: alpha * W2 (D, FF)
    const float *bias;     // ROWLN / GLU / QKV: (N); FFN: b1 (FF)
    const float *bias2;    // FFN: b2 (D)
    int N;                 // output columns (D / 2D / 3D) or FF
    int K;                 // FRONT: depth of the product (F C of the frontend's output linear, a multiple of 256); W is (D, K), its operand ChainArgs::A0 is (M, K)
    float alpha;           // FFN residual factor (already folded into W2; applied to b2 here)
    const float *g1, *b1, *g2, *b2;    // LayerNorm(s) after the residual add (g2 != null: chained, x <- LN1)
    int store_x, store_xn;             // write the fp32 stream / the normalised operand back to global after this stage
    bf16_t *out;           // GLU: (M, D)
    bf16_t *q, *k, *v;     // QKV
    float *tap_pre, *tap_post;         // debug taps (TAPS instantiation only): (M, D) fp32 copies of the stream after this stage's residual
                                       // add, and (chained LayerNorms) after the first LayerNorm; null = not wanted
};

struct ChainArgs {
    const bf16_t *A0;      // first operand rows (M, D) (unused with the depthwise prologue); (M, st[0].K) when the first stage is FRONT
    float *x;              // fp32 residual stream (M, D): read at the start, written by the stages that have store_x
    int x_in_blocked, x_out_blocked;   // the stream in the kernels' own register order instead of row-major: [row block][wave][row tile][column
                                       // tile][lane] float4, i.e. every wave-instruction moves 1 KB of consecutive bytes (row-major, a wave's
                                       // 16 rows x 4 lanes x 16 bytes are sixteen 64-byte pieces).  Producer and consumer must use the same
                                       // rows per workgroup; the buffer holds whole row blocks (cocr_api: workspace).
    bf16_t *xn;            // normalised operand (M, D), written when a stage asks for it
    int M, nstages;
    int kd, kl;            // zero-padded narrow models: k-steps (of 32) with real columns in a K = D product / in the FFN's last hidden chunk (8 = all)
    int xcd_order;         // row blocks in XCD-contiguous order (rowchain.hip.h); every launch of a forward uses the same setting (blocked stream layout)
    const bf16_t *dw_in;   // depthwise-conv prologue: GLU output (M, D)
    const float *dw_w, *dw_b;      // BatchNorm-folded depthwise taps [k][D] and bias [D]
    bf16_t *tap_dw;        // debug tap (TAPS instantiation): (M, D) copy of the prologue's output, or null
    int dh, dhp, heads, T_, Tp;       // attention layout of the QKV stage
    float inv_d;           // 1 / (LayerNorm width): 1 / D, or 1 / (the model's own encoder_dim) when D is a zero-padded width (cocr_api: set_engine_dims)
    unsigned long long *stamps;    // dev builds (COCR_CHAIN_STAMPS_BUILD): host-visible cycle stamps [wave][64] of workgroup 7, or null
    ChainStage st[4];
};

static inline bool rowchain_supported(int D, int FF, int dh) { return (D == 256 || D == 512) && FF % 256 == 0 && FF >= 256 && FF <= 2048 && dh % 8 == 0 && D % dh == 0; }

// `taps`: the debug instantiation; `rows_hint`: rows per workgroup (0 = pick by the number of rows, see rowchain_pick_mt)
hipError_t launch_rowchain_256(hipStream_t s, const ChainArgs &a, bool taps, int rows_hint);
hipError_t launch_rowchain_512(hipStream_t s, const ChainArgs &a, bool taps, int rows_hint);
