// C[M,N] = A[M,K] . W[N,K]^T with fused epilogues -- every dense contraction of the path:
// frontend pointwise conv F3 and flatten-linear F5 (reference conformer/convolution.py:207-213,224),
// FFN linears (feed_forward.py:47,50), q/k/v/out projections (attention.py:59-61,70), the conv
// module's pointwise convs (convolution.py:138,143), the positional projection (attention.py:62) and
// the decoder (pred.py:90).  Both operands are k-contiguous, which is the MFMA fragment order
// (common.hip.h), so activations (M,K) and nn.Linear weights (N,K) are used as stored.
//
// Two kernels share one epilogue:
//   gemm_ring_kernel   K % BK == 0.  Operand tiles go global -> LDS by LDS-DMA (global_load_lds, 16 B per
//                      lane, no VGPR staging) into an NST-deep ring; the k-loop waits with a COUNTED vmcnt so
//                      NST-1 tiles stay in flight across the raw barrier (one barrier per k-tile).  The LDS
//                      image is lane-linear (one DMA wave-instruction = 1 KiB = 8 rows of 128 B); the
//                      bank-conflict swizzle sits on the per-lane SOURCE address and is undone at the fragment
//                      read: LDS chunk position p of row r holds global chunk p ^ (r & 7).
//   gemm_stream_kernel any K % (16/sizeof(T)) == 0.  Register-staged tiles (zero-filled k tail), two buffers.
// Both: 256 threads = 4 waves as 2x2 over a BMxBN tile; k-tile = 128 bytes per row; XCD-aware tile order;
// MFMA issued with the operands swapped (weights on the row side) so that a lane holds 4 consecutive output
// columns of one row; the epilogue applies bias / activation / GLU in registers, stages the tile in LDS and
// writes FULL ROWS, 16 bytes per lane, consecutive lanes consecutive addresses.  (Measured on MI355X: the
// direct 8-byte-per-lane store of the accumulator layout -- 32-byte row fragments -- took 2/3 of the kernel.)
#pragma once
#include <algorithm>
#include <set>
#include <type_traits>

#include "common.hip.h"

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2 };

// XCD-aware workgroup remap (8 XCDs, private L2 each; workgroups are dealt round-robin, so ids b and b+8
// share an XCD).  Logical tile L = chunk(b % 8) + b / 8 gives every XCD a CONTIGUOUS run of logical tiles:
// all column tiles of a row block then read that row block's A rows through ONE L2 instead of eight.
// Bijective for any grid size (speed only, never correctness).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// ---- epilogues -------------------------------------------------------------------------------------------
// stage_t            element type of the LDS-staged tile (compute dtype or float)
// transform(n, v, r) 4 accumulator columns n..n+3 (n % 4 == 0) -> 4 staged values; GLU: (n, value, gate, r)
// store(m, c, src, cnt)  `cnt` staged values of row m starting at staged column c (c % CH == 0, CH = 16/sizeof
//                    (stage_t)); src points into LDS, 16-byte aligned; cnt == CH except at the right edge.
template <typename U> __device__ __forceinline__ void copy16(U *dst, const U *src) {
    *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(src);
}

// out[m, n] = act(acc + bias[n]) in the compute dtype (hidden activations)
template <typename T, int ACT> struct EpiBiasAct {
    typedef T stage_t;
    static constexpr bool GLU = false;
    static constexpr bool ROWWISE = false;
    T *out; int ldo; const float *bias; int N;
    __device__ __forceinline__ void transform(int n, const float *v, float *r) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float x = v[i] + (n + i < N ? bias[n + i] : 0.f);
            if (ACT == ACT_RELU) x = fmaxf(x, 0.0f);
            if (ACT == ACT_SILU) x = silu_f(x);
            r[i] = x;
        }
    }
    __device__ __forceinline__ void store(int m, int c, const T *src, int cnt) const {
        T *p = out + (size_t)m * ldo + c;
        if (cnt == 16 / (int)sizeof(T) && (ldo % (16 / (int)sizeof(T))) == 0) copy16(p, src);
        else for (int i = 0; i < cnt; ++i) p[i] = src[i];
    }
};

// out[m, n] = acc + bias[n] as fp32 (frontend output = residual stream; decoder logits)
struct EpiStoreF32 {
    typedef float stage_t;
    static constexpr bool GLU = false;
    static constexpr bool ROWWISE = false;
    float *out; int ldo; const float *bias; int N;     // bias may be null (split-K partial sums)
    size_t zstride = 0;                                // split-K: partial z goes to out + z * zstride
    __device__ __forceinline__ EpiStoreF32 with_z(int z) const { EpiStoreF32 e = *this; e.out += (size_t)z * zstride; return e; }
    __device__ __forceinline__ void transform(int n, const float *v, float *r) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = v[i] + ((bias && n + i < N) ? bias[n + i] : 0.f);
    }
    __device__ __forceinline__ void store(int m, int c, const float *src, int cnt) const {
        float *p = out + (size_t)m * ldo + c;
        if (cnt == 4 && (ldo & 3) == 0) copy16(p, src);
        else for (int i = 0; i < cnt; ++i) p[i] = src[i];
    }
};

// Decoder logits + the greedy decoder's first half (kraken greedy_decoder: argmax over classes per frame, first index on ties -- the call
// sites pred.py:143,162,177): out[m, :] = acc + bias as fp32, flab[m] = argmax, fval[m] = max.  Row-complete (BN >= N <= 128): a 16-lane
// group takes one staged row, lane c of it the float4 chunks c and c + 16 (columns ascending inside a lane: strict > keeps the lowest
// index), then four DPP steps inside the 16-lane row on (value, index) pairs, equal values resolved towards the lower index.
struct EpiLogitsArgmax {
    typedef float stage_t;
    static constexpr bool GLU = false;
    static constexpr bool ROWWISE = true;
    float *out; int ldo; const float *bias; int N;
    int32_t *flab; float *fval;
    __device__ __forceinline__ void transform(int n, const float *v, float *r) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = v[i] + (n + i < N ? bias[n + i] : 0.f);
    }
    template <int CTRL> static __device__ __forceinline__ void step(float &v, int &i) {
        const float ov = dpp_f32<CTRL, 0xF>(v, v);
        const int oi = __builtin_amdgcn_update_dpp(i, i, CTRL, 0xF, 0xF, false);
        const bool take = ov > v || (ov == v && oi < i);
        v = take ? ov : v;
        i = take ? oi : i;
    }
    __device__ __forceinline__ void rows4(int m0, int M, const float *staged, int rs_floats, int lane) const {
        const int rl = lane >> 4, cl = lane & 15, m = m0 + rl;
        const float *row = staged + rl * rs_floats;
        float best = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = 4 * (cl + 16 * h);
            if (c < N) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(row + c);
                if (m < M) {
                    if (c + 3 < N && (ldo & 3) == 0) *reinterpret_cast<f32x4 *>(out + (size_t)m * ldo + c) = v;
                    else for (int i = 0; i < 4 && c + i < N; ++i) out[(size_t)m * ldo + c + i] = v[i];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (c + i < N && (v[i] > best || bi == 0x7fffffff)) { best = v[i]; bi = c + i; }
            }
        }
        step<0xB1>(best, bi);          // quad_perm(1,0,3,2)
        step<0x4E>(best, bi);          // quad_perm(2,3,0,1)
        step<0x141>(best, bi);         // row_half_mirror
        step<0x140>(best, bi);         // row_mirror
        if (cl == 0 && m < M) { flab[m] = bi; fval[m] = best; }
    }
};

// x[m, n] += alpha * (acc + bias[n])  -- ResidualConnectionModule, modules.py:32 (input_factor 1)
struct EpiResidual {
    typedef float stage_t;
    static constexpr bool GLU = false;
    static constexpr bool ROWWISE = false;
    float *x; int ldx; const float *bias; float alpha; int N;
    __device__ __forceinline__ void transform(int n, const float *v, float *r) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = alpha * (v[i] + (n + i < N ? bias[n + i] : 0.f));
    }
    __device__ __forceinline__ void store(int m, int c, const float *src, int cnt) const {
        float *p = x + (size_t)m * ldx + c;
        if (cnt == 4 && (ldx & 3) == 0) {
            f32x4 o = *reinterpret_cast<f32x4 *>(p);
            const f32x4 d = *reinterpret_cast<const f32x4 *>(src);
            o += d;
            *reinterpret_cast<f32x4 *>(p) = o;
        } else {
            for (int i = 0; i < cnt; ++i) p[i] += src[i];
        }
    }
};

// GLU over channels (convolution.py:139): the weight rows are interleaved at load time so that the
// 16-column tile 2j holds the value half of channels 16j..16j+15 and tile 2j+1 their gates:
// out[m, 16j + c] = (a + ba) * sigmoid(g + bg).  N (= 2D) is a multiple of 32.
template <typename T> struct EpiGLU {
    typedef T stage_t;
    static constexpr bool GLU = true;
    static constexpr bool ROWWISE = false;
    T *out; int ldo; const float *bias; int N;
    __device__ __forceinline__ void transform(int n, const float *v, const float *gte, float *r) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = (v[i] + bias[n + i]) * sigmoid_f(gte[i] + bias[n + 16 + i]);
    }
    __device__ __forceinline__ void store(int m, int c, const T *src, int cnt) const {
        T *p = out + (size_t)m * ldo + c;
        if (cnt == 16 / (int)sizeof(T)) copy16(p, src);
        else for (int i = 0; i < cnt; ++i) p[i] = src[i];
    }
};

// fused q/k/v projection (N = 3D): rows scattered into the attention layout [B][h][Tp][dhp] of q, k, v
// (head dim padded to a multiple of 32 with zeros; rows >= T unused).
template <typename T> struct EpiQKV {
    typedef T stage_t;
    static constexpr bool GLU = false;
    static constexpr bool ROWWISE = false;
    T *q, *k, *v; const float *bias; int D, dh, dhp, heads, T_, Tp, N;
    __device__ __forceinline__ void transform(int n, const float *a, float *r) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = a[i] + (n + i < N ? bias[n + i] : 0.f);
    }
    __device__ __forceinline__ T *dst(int b, int t, int n) const {
        const int which = n / D, hd = n - which * D, hh = hd / dh, d = hd - hh * dh;
        T *base = which == 0 ? q : (which == 1 ? k : v);
        return base + (((size_t)b * heads + hh) * Tp + t) * dhp + d;
    }
    __device__ __forceinline__ void store(int m, int c, const T *src, int cnt) const {
        const int b = m / T_, t = m - b * T_;
        constexpr int CH = 16 / (int)sizeof(T);
        if (cnt == CH && (dh % CH) == 0 && (D % CH) == 0) copy16(dst(b, t, c), src);   // the chunk stays inside one head
        else for (int i = 0; i < cnt; ++i) *dst(b, t, c + i) = src[i];
    }
};

// positional projection table P = PE Wpos^T into [row][h][dhp] (head-padded)
template <typename T> struct EpiPosTable {
    typedef T stage_t;
    static constexpr bool GLU = false;
    static constexpr bool ROWWISE = false;
    T *out; int dh, dhp, heads;
    __device__ __forceinline__ void transform(int n, const float *v, float *r) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = v[i];
    }
    __device__ __forceinline__ void store(int m, int c, const T *src, int cnt) const {
        for (int i = 0; i < cnt; ++i) {
            const int hh = (c + i) / dh, d = (c + i) - hh * dh;
            out[((size_t)m * heads + hh) * dhp + d] = src[i];
        }
    }
};

// Row-complete epilogue for the N == encoder_dim products (frontend output linear, FFN down, attention out,
// conv-module pointwise 2): requires BN >= N so that a workgroup owns whole rows.  After the accumulator tile
// (alpha * (acc + bias)) is staged in LDS, each wave takes 4 rows at a time and, with the row in registers,
//   x_new = x + staged                      (ResidualConnectionModule, modules.py:32; no `x +` for the frontend)
//   single:  x <- x_new ;            xn <- LN1(x_new)                 (the next module's LayerNorm)
//   chained: x <- LN1(x_new) ;       xn <- LN2(LN1(x_new))            (block-final LayerNorm encoder.py:99 + next FFN's)
// so no standalone LayerNorm launch and no re-read of the stream is left in a block.
template <typename T, int NV> struct EpiResidualLN {
    typedef float stage_t;
    static constexpr bool GLU = false;
    static constexpr bool ROWWISE = true;
    float *x; int D; const float *bias; float alpha; int N; int has_resid;
    const float *g1, *b1, *g2, *b2;
    T *xn;
    int Dn = 0;            // LayerNorm width when D is a zero-padded row width (0: D).  Columns [Dn, D) hold exact zeros: the sums need no mask, the
                           // centred squares and the divisor do
    __device__ __forceinline__ void transform(int n, const float *v, float *r) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = alpha * (v[i] + (n + i < N ? bias[n + i] : 0.f));
    }
    // The residual rows can be fetched long before the accumulators are ready (kernel prologue): the epilogue then has no
    // global-load latency left in it.  rows m0 .. m0+3; clamped addresses, rows >= M are discarded at the stores.
    struct Rows4 { f32x4 v[4][NV]; };
    __device__ __forceinline__ Rows4 rows4_load(int m0, int M, int lane) const {
        const int nchunk = D >> 2;
        Rows4 p;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int cc = min(lane + 64 * v, nchunk - 1);
                p.v[r][v] = *reinterpret_cast<const f32x4 *>(x + (size_t)min(m0 + r, M - 1) * D + 4 * cc);
            }
        return p;
    }
    __device__ __forceinline__ void rows4(int m0, int M, const float *staged, int rs_floats, int lane) const {
        rows4(m0, M, staged, rs_floats, lane, rows4_load(m0, M, lane));
    }
    // Chain form (chain.hip.h): the residual rows come in through `xr` and the NEW stream rows go back out through it (the next
    // stage of the chain adds to them without touching global memory); x / xn are written to global only when asked; the new
    // normalised rows are also written as bf16 into the swizzled operand image `lds` (panels of [rows][128 B], tile row = row0 + r).
    __device__ __forceinline__ void rows4_chain(int m0, int M, const float *staged, int rs_floats, int lane, Rows4 &xr, int store_x, int store_xn,
                                                unsigned char *lds, int row0, int panel_bytes) const {
        static_assert(NV == 1, "chain kernels are written for encoder_dim <= 256");
        const int nchunk = D >> 2;
        const float inv_d = 1.0f / (float)D;
        const int c = lane, cc = min(c, nchunk - 1);
        f32x4 xv[4];
        float s[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f32x4 t = *reinterpret_cast<const f32x4 *>(staged + r * rs_floats + 4 * cc);
            if (has_resid) t += xr.v[r][0];
            if (c >= nchunk) t = (f32x4){0, 0, 0, 0};
            xv[r] = t;
            s[r] = t[0] + t[1] + t[2] + t[3];
        }
        auto normalise = [&](const float *gam, const float *bet) {
            float mean[4], q[4], rstd[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) mean[r] = wave_sum(s[r]) * inv_d;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                q[r] = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = c < nchunk ? xv[r][e] - mean[r] : 0.f; q[r] += d * d; }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) rstd[r] = 1.0f / sqrtf(wave_sum(q[r]) * inv_d + 1e-5f);
            const f32x4 ga = c < nchunk ? *reinterpret_cast<const f32x4 *>(gam + 4 * cc) : (f32x4){0, 0, 0, 0};
            const f32x4 be = c < nchunk ? *reinterpret_cast<const f32x4 *>(bet + 4 * cc) : (f32x4){0, 0, 0, 0};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int e = 0; e < 4; ++e) xv[r][e] = (xv[r][e] - mean[r]) * rstd[r] * ga[e] + be[e];
                s[r] = xv[r][0] + xv[r][1] + xv[r][2] + xv[r][3];
            }
        };
        auto keep_x = [&]() {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xr.v[r][0] = xv[r];
                if (store_x && m0 + r < M && c < nchunk) *reinterpret_cast<f32x4 *>(x + (size_t)(m0 + r) * D + 4 * c) = xv[r];
            }
        };
        if (!g2) keep_x();
        normalise(g1, b1);
        if (g2) { keep_x(); normalise(g2, b2); }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            bf16x4 o = {(T)xv[r][0], (T)xv[r][1], (T)xv[r][2], (T)xv[r][3]};
            if (c < nchunk) {
                const int row = row0 + r, col = 4 * c;
                *reinterpret_cast<bf16x4 *>(lds + (col >> 6) * panel_bytes + row * 128 + ((((col & 63) >> 3) ^ (row & 7)) << 4) + ((col & 7) >> 2) * 8) = o;
                if (store_xn && m0 + r < M) *reinterpret_cast<bf16x4 *>(xn + (size_t)(m0 + r) * D + col) = o;
            }
        }
    }
    // rows m0 .. m0+3 (row r valid if m0 + r < M); staged row r at `staged + r * rs_floats`; xr = rows4_load(m0, M, lane)
    __device__ __forceinline__ void rows4(int m0, int M, const float *staged, int rs_floats, int lane, const Rows4 &xr) const {
        const int nchunk = D >> 2, nnorm = (Dn > 0 ? Dn : D) >> 2;
        const float inv_d = 1.0f / (float)(Dn > 0 ? Dn : D);
        f32x4 xv[4][NV];
        float s[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s[r] = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int c = lane + 64 * v;
                const int cc = min(c, nchunk - 1);
                f32x4 t = *reinterpret_cast<const f32x4 *>(staged + r * rs_floats + 4 * cc);
                if (has_resid) t += xr.v[r][v];
                if (c >= nchunk) t = (f32x4){0, 0, 0, 0};
                xv[r][v] = t;
                s[r] += t[0] + t[1] + t[2] + t[3];
            }
        }
        auto write_x = [&]() {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const int c = lane + 64 * v;
                    if (m0 + r < M && c < nchunk) *reinterpret_cast<f32x4 *>(x + (size_t)(m0 + r) * D + 4 * c) = xv[r][v];
                }
        };
        auto normalise = [&](const float *gam, const float *bet) {
            float mean[4], q[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) mean[r] = wave_sum(s[r]) * inv_d;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                q[r] = 0.f;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const bool ok = lane + 64 * v < nnorm;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = ok ? xv[r][v][e] - mean[r] : 0.f; q[r] += d * d; }
                }
            }
            float rstd[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) rstd[r] = 1.0f / sqrtf(wave_sum(q[r]) * inv_d + 1e-5f);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int c = lane + 64 * v;
                const bool ok = c < nchunk;
                const f32x4 ga = ok ? *reinterpret_cast<const f32x4 *>(gam + 4 * c) : (f32x4){0, 0, 0, 0};
                const f32x4 be = ok ? *reinterpret_cast<const f32x4 *>(bet + 4 * c) : (f32x4){0, 0, 0, 0};
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) xv[r][v][e] = (xv[r][v][e] - mean[r]) * rstd[r] * ga[e] + be[e];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[r] = 0.f;
#pragma unroll
                for (int v = 0; v < NV; ++v) s[r] += xv[r][v][0] + xv[r][v][1] + xv[r][v][2] + xv[r][v][3];
            }
        };
        if (!g2) write_x();
        normalise(g1, b1);
        if (g2) { write_x(); normalise(g2, b2); }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int c = lane + 64 * v;
                if (m0 + r < M && c < nchunk) {
                    T *p = xn + (size_t)(m0 + r) * D + 4 * c;
                    if constexpr (sizeof(T) == 2) { bf16x4 o = {(T)xv[r][v][0], (T)xv[r][v][1], (T)xv[r][v][2], (T)xv[r][v][3]}; *reinterpret_cast<bf16x4 *>(p) = o; }
                    else { *reinterpret_cast<f32x4 *>(p) = xv[r][v]; }
                }
            }
    }
};

template <typename S> __device__ __forceinline__ void stage4(S *dst, const float *r);
template <> __device__ __forceinline__ void stage4<float>(float *dst, const float *r) { *reinterpret_cast<f32x4 *>(dst) = (f32x4){r[0], r[1], r[2], r[3]}; }
template <> __device__ __forceinline__ void stage4<bf16_t>(bf16_t *dst, const float *r) {
    bf16x4 o = {(bf16_t)r[0], (bf16_t)r[1], (bf16_t)r[2], (bf16_t)r[3]};
    *reinterpret_cast<bf16x4 *>(dst) = o;
}

// bytes of LDS the staged epilogue needs
template <typename Epi, int BM, int BN> constexpr size_t epi_lds_bytes() {
    return (size_t)BM * ((Epi::GLU ? BN / 2 : BN) * sizeof(typename Epi::stage_t) + 16);
}

// Accumulator (i, j), reg q of a wave = element (m = 16 i + r16, n = 16 j + 4 g + q) of the wave's (BM/2)x(BN/2)
// quadrant.  Must be entered by all 256 threads after the last LDS read of the k-loop has been fenced by a barrier.
template <int BM, int BN, int MI, int NI, typename Epi>
__device__ __forceinline__ void gemm_epilogue(f32x4 (&acc)[MI][NI], unsigned char *smem, int m0, int n0, int M, int N, const Epi &epi) {
    typedef typename Epi::stage_t S;
    constexpr int COLS = Epi::GLU ? BN / 2 : BN;
    constexpr int RS = COLS * (int)sizeof(S) + 16;          // staged row stride, bytes
    constexpr int CH = 16 / (int)sizeof(S);                 // elements per 16-byte chunk
    constexpr int CPR = COLS / CH;                          // chunks per staged row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4, wm = wave >> 1, wn = wave & 1;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int row = wm * (BM / 2) + 16 * i + r16;
        if constexpr (Epi::GLU) {
#pragma unroll
            for (int j = 0; j < NI; j += 2) {
                const int nl = wn * (BN / 2) + 16 * j + 4 * g;          // local column of the value tile
                const float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                const float w[4] = {acc[i][j + 1][0], acc[i][j + 1][1], acc[i][j + 1][2], acc[i][j + 1][3]};
                float r[4] = {0.f, 0.f, 0.f, 0.f};
                if (n0 + nl + 19 < N) epi.transform(n0 + nl, v, w, r);
                stage4<S>(reinterpret_cast<S *>(smem + row * RS) + ((nl >> 5) << 4) + (nl & 15), r);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int nl = wn * (BN / 2) + 16 * j + 4 * g;
                const float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                float r[4];
                epi.transform(n0 + nl, v, r);
                stage4<S>(reinterpret_cast<S *>(smem + row * RS) + nl, r);
            }
        }
    }
    __syncthreads();
    if constexpr (Epi::ROWWISE) {      // BN >= N: the tile holds complete rows (n0 == 0)
        static_assert(BM % 16 == 0, "4 waves x 4 rows");
        for (int rr = wave * 4; rr < BM; rr += 16)
            epi.rows4(m0 + rr, M, reinterpret_cast<const float *>(smem + rr * RS), RS / 4, lane);
    } else {
        const int c0 = Epi::GLU ? n0 / 2 : n0, cend = Epi::GLU ? N / 2 : N;     // staged-column range of this tile in the output
        for (int id = threadIdx.x; id < BM * CPR; id += 256) {
            const int row = id / CPR, ch = id - row * CPR;
            const int m = m0 + row, c = c0 + ch * CH;
            if (m < M && c < cend) epi.store(m, c, reinterpret_cast<const S *>(smem + row * RS) + ch * CH, min(CH, cend - c));
        }
    }
}

// ---- kernel arguments ----------------------------------------------------------------------------
template <typename T> struct GemmArgs {
    const T *A; int lda;
    const T *W; int ldw;
    int M, N, K;           // K of ONE split
    int k_zstride;         // split-K: grid.y = number of splits; split z reads k in [z * k_zstride, z * k_zstride + K) and
                           // writes through epi.with_z(z) (0 when unused)
    // batched products (gemm_ring_kernel with EpiStoreF32; grid.y = batches): batch z = (zb, zh) = (z / z_div, z % z_div) reads A and W
    // at element offsets zb * ?_zb + zh * ?_zh and writes at out + zb * o_zb + zh * o_zh -- the (line, head) sub-matrices of (M, D)
    // activations and per-(line, head) square matrices alike.  z_div = 0: not batched.
    int z_div = 0;
    long long a_zb = 0, a_zh = 0, w_zb = 0, w_zh = 0, o_zb = 0, o_zh = 0;
};

__device__ __forceinline__ bf16x8 lds_frag_swz(const unsigned char *row, int kc, int g, int swz, bf16_t) {
    return *reinterpret_cast<const bf16x8 *>(row + (((kc * 4 + g) ^ swz) << 4));
}
__device__ __forceinline__ f32x8 lds_frag_swz(const unsigned char *row, int kc, int g, int swz, float) {
    const f32x4 lo = *reinterpret_cast<const f32x4 *>(row + (((2 * g) ^ swz) << 4));
    const f32x4 hi = *reinterpret_cast<const f32x4 *>(row + (((2 * g + 1) ^ swz) << 4));
    return (f32x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---- LDS-DMA ring kernel (K % BK == 0) --------------------------------------------------------------------
template <typename T, int BM, int BN, int NST, typename Epi>
__global__ __launch_bounds__(256) void gemm_ring_kernel(GemmArgs<T> p, Epi epi) {
    constexpr int ROWB = 128;
    constexpr int BK = ROWB / sizeof(T);
    constexpr int MI = BM / 32, NI = BN / 32;
    constexpr int PER = BM / 32 + BN / 32;                 // DMA wave-instructions per wave per k-tile
    constexpr int STAGE = (BM + BN) * ROWB;
    static_assert(NST >= 2 && NST <= 4 && BM % 32 == 0 && BN % 64 == 0, "config");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int M = p.M, N = p.N, K = p.K;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4, swz = r16 & 7;
    const int wm = wave >> 1, wn = wave & 1;
    const int gx = (N + BN - 1) / BN;
    const int ltile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (ltile / gx) * BM, n0 = (ltile % gx) * BN;
    const int nk = K / BK;
    if (p.k_zstride) { p.A += (size_t)blockIdx.y * p.k_zstride; p.W += (size_t)blockIdx.y * p.k_zstride; }
    long long out_off = 0;
    if (p.z_div) {
        const int z = blockIdx.y, zb = z / p.z_div, zh = z - zb * p.z_div;
        p.A += zb * p.a_zb + zh * p.a_zh;
        p.W += zb * p.w_zb + zh * p.w_zh;
        out_off = zb * p.o_zb + zh * p.o_zh;
    }

    // Per-lane byte offsets of this thread's DMA pieces are loop-invariant (row clamp + swizzled chunk); a k-tile only moves the
    // UNIFORM base by 128 bytes, so an issue costs no vector arithmetic (it was ~60 VALU instructions per k-tile beside 32 MFMAs).
    // 32-bit offsets: the launchers check (rows - 1) * ld * sizeof(T) + 128 < 4 GiB.
    unsigned offA[BM / 32], offW[BN / 32];
    {
        const int lrow = lane >> 3, cpos = lane & 7;
#pragma unroll
        for (int i = 0; i < BM / 32; ++i) {
            const int row = (wave + 4 * i) * 8 + lrow;
            offA[i] = (unsigned)min(m0 + row, M - 1) * (unsigned)(p.lda * (int)sizeof(T)) + ((cpos ^ (row & 7)) << 4);
        }
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) {
            const int row = (wave + 4 * i) * 8 + lrow;
            offW[i] = (unsigned)min(n0 + row, N - 1) * (unsigned)(p.ldw * (int)sizeof(T)) + ((cpos ^ (row & 7)) << 4);
        }
    }
    auto issue = [&](int kt) {
        unsigned char *st = smem + (kt % NST) * STAGE;
        const char *abase = reinterpret_cast<const char *>(p.A) + (size_t)kt * ROWB;      // uniform
        const char *wbase = reinterpret_cast<const char *>(p.W) + (size_t)kt * ROWB;
#pragma unroll
        for (int i = 0; i < BM / 32; ++i)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(abase + offA[i]), (lds_ptr_t)(st + (wave + 4 * i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < BN / 32; ++i)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wbase + offW[i]), (lds_ptr_t)(st + BM * ROWB + (wave + 4 * i) * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < nk) issue(t);

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < nk; ++kt) {
        // tile kt must have landed: only tiles issued after it may still be in flight
        const int after = min(nk, kt + NST - 1) - (kt + 1);
        if (after >= 2) wait_vmcnt<2 * PER>(); else if (after == 1) wait_vmcnt<PER>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + NST - 1 < nk) issue(kt + NST - 1);       // refills the slot every wave finished reading before this barrier
        const unsigned char *st = smem + (kt % NST) * STAGE;
        const unsigned char *sa = st + (wm * (BM / 2) + r16) * ROWB;
        const unsigned char *sw = st + BM * ROWB + (wn * (BN / 2) + r16) * ROWB;
#pragma unroll
        for (int kc = 0; kc < BK / 32; ++kc) {
            typename FragOf<T>::type a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = lds_frag_swz(sa + i * 16 * ROWB, kc, g, swz, T());
#pragma unroll
            for (int j = 0; j < NI; ++j) b[j] = lds_frag_swz(sw + j * 16 * ROWB, kc, g, swz, T());
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = mma16(b[j], a[i], acc[i][j]);   // swapped: rows = n, columns = m
        }
    }
    __syncthreads();                                     // every wave is done reading the ring: reuse it for the staged tile
    if constexpr (std::is_same<Epi, EpiStoreF32>::value) {
        EpiStoreF32 ez = epi.with_z(p.k_zstride ? blockIdx.y : 0);
        ez.out += out_off;
        gemm_epilogue<BM, BN, MI, NI, Epi>(acc, smem, m0, n0, M, N, ez);
    } else {
        gemm_epilogue<BM, BN, MI, NI, Epi>(acc, smem, m0, n0, M, N, epi);
    }
}

// ---- both operands K-MAJOR (round 4; the weight-gradient product of a training step: dW = dY^T X with dY (rows, N_out) and X (rows, K_in)
// as the forward left them) ---------------------------------------------------------------------------------------------------------------
//     out (M, N) = sum_k A[k][m] B[k][n]        A: (K, M) row-major, lda;  B: (K, N) row-major, ldb;  bf16, fp32 accumulation
// No transposed copies: a k-tile of 32 rows x 128 columns of each operand goes to LDS as it lies in memory (LDS-DMA, 256-byte rows, the
// 16-byte chunk c of row r stored at chunk c ^ 2 (r & 7)), and the MFMA fragments -- 8 consecutive k of one column per lane -- are read
// with ds_read_b64_tr_b16 (two per fragment: rows 4 g + q and 16 + 4 g + q; both operands see the same permutation of k inside a chunk,
// which a product does not notice).  The swizzle puts the 8 rows a 32-lane half touches (32 bytes each) on all 64 banks.
// Requirements: M % 8 == 0, N % 8 == 0, K % 32 == 0 (the callers pad the rows with zeros), 16-byte aligned bases and leading dimensions.
template <int BM, int BN, int NST>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmArgs<bf16_t> p, EpiStoreF32 epi) {
    typedef bf16_t T;
    constexpr int BK = 32, MI = BM / 32, NI = BN / 32;
    constexpr int AROW = BM * 2, BROW = BN * 2;              // bytes per LDS row
    constexpr int STAGE = BK * (AROW + BROW);
    constexpr int PA = BK * AROW / 1024 / 4, PB = BK * BROW / 1024 / 4;      // DMA wave-instructions per wave and k-tile
    static_assert(BM == 128 && BN == 128, "one 1 KiB wave-instruction = 4 rows of 256 bytes");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int M = p.M, N = p.N, K = p.K;                     // K: this launch's (split's) depth
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int gx = (N + BN - 1) / BN;
    const int ltile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (ltile / gx) * BM, n0 = (ltile % gx) * BN;
    const int nk = K / BK;
    const size_t kz = (size_t)blockIdx.y * p.k_zstride;      // split-K: this split's first k row
    const unsigned char *Ab = reinterpret_cast<const unsigned char *>(p.A + kz * p.lda), *Bb = reinterpret_cast<const unsigned char *>(p.W + kz * p.ldw);
    // DMA pieces of this lane: LDS slot (row r = 4 instr + (lane >> 4), chunk s = lane & 15) <- global chunk s ^ 2 (r & 7) of that row
    // (clamped to the operand's last chunk: columns beyond M / N feed only outputs nobody stores)
    unsigned offA[PA], offB[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int r = 4 * (wave + 4 * i) + (lane >> 4), c = (lane & 15) ^ (2 * (r & 7));
        offA[i] = (unsigned)r * (unsigned)(p.lda * 2) + (unsigned)min(m0 + 8 * c, M - 8) * 2u;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int r = 4 * (wave + 4 * i) + (lane >> 4), c = (lane & 15) ^ (2 * (r & 7));
        offB[i] = (unsigned)r * (unsigned)(p.ldw * 2) + (unsigned)min(n0 + 8 * c, N - 8) * 2u;
    }
    auto issue = [&](int kt) {
        unsigned char *st = smem + (kt % NST) * STAGE;
        const unsigned char *ab = Ab + (size_t)kt * BK * p.lda * 2, *bb = Bb + (size_t)kt * BK * p.ldw * 2;      // uniform
#pragma unroll
        for (int i = 0; i < PA; ++i) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(ab + offA[i]), (lds_ptr_t)(st + (wave + 4 * i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < PB; ++i) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(bb + offB[i]), (lds_ptr_t)(st + BK * AROW + (wave + 4 * i) * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < nk) issue(t);
    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // transposed fragment of the 16 columns from `col0` of a [BK][cols] tile: lane 16 g + 4 q + pp reads 8 bytes of row 4 g + q (and 16 rows
    // on), columns col0 + 4 pp .. + 3, and receives column (lane & 15) of the four rows
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    auto tr_frag = [&](const unsigned char *tile, int rowb, int col0) {
        typedef __attribute__((address_space(3))) bf16x4 *lp;
        const int r = 4 * g + q, ch = (col0 >> 3) + (pp >> 1);
        const unsigned char *p0 = tile + r * rowb + ((ch ^ (2 * (r & 7))) << 4) + 8 * (pp & 1);
        const unsigned char *p1 = tile + (r + 16) * rowb + ((ch ^ (2 * ((r + 16) & 7))) << 4) + 8 * (pp & 1);
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)p0), hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)p1);
        return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    for (int kt = 0; kt < nk; ++kt) {
        const int after = min(nk, kt + NST - 1) - (kt + 1);
        if (after >= 2) wait_vmcnt<2 * (PA + PB)>(); else if (after == 1) wait_vmcnt<PA + PB>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + NST - 1 < nk) issue(kt + NST - 1);
        const unsigned char *sa = smem + (kt % NST) * STAGE, *sb = sa + BK * AROW;
        bf16x8 a[MI], b[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i] = tr_frag(sa, AROW, wm * (BM / 2) + 16 * i);
#pragma unroll
        for (int j = 0; j < NI; ++j) b[j] = tr_frag(sb, BROW, wn * (BN / 2) + 16 * j);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = mma16(b[j], a[i], acc[i][j]);   // swapped: rows = n, columns = m
    }
    __syncthreads();
    EpiStoreF32 ez = epi.with_z(p.k_zstride ? blockIdx.y : 0);
    gemm_epilogue<BM, BN, MI, NI, EpiStoreF32>(acc, smem, m0, n0, M, N, ez);
}

// ---- register-staged kernel (any k tail) ---------------------------------------------------------------------
template <typename T, int BM, int BN, typename Epi>
__global__ __launch_bounds__(256) void gemm_stream_kernel(GemmArgs<T> p, Epi epi) {
    constexpr int ROWB = 128;
    constexpr int BK = ROWB / sizeof(T);
    constexpr int EPC = 16 / sizeof(T);
    constexpr int STRIDE = ROWB + 16;
    constexpr int MI = BM / 32, NI = BN / 32;
    constexpr int A_IT = BM * 8 / 256, W_IT = BN * 8 / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int M = p.M, N = p.N, K = p.K;
    const int Kp = (K + 31) & ~31;
    unsigned char *a_area = smem, *w_area = smem + 2 * BM * STRIDE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int gx = (N + BN - 1) / BN;
    const int ltile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (ltile / gx) * BM, n0 = (ltile % gx) * BN;

    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 ra[A_IT], rw[W_IT];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int id = it * 256 + tid, row = id >> 3, ch = id & 7;
            const int m = m0 + row, k = k0 + ch * EPC;
            ra[it] = (m < M && k < K) ? *reinterpret_cast<const u32x4 *>(p.A + (size_t)m * p.lda + k) : (u32x4){0, 0, 0, 0};
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int id = it * 256 + tid, row = id >> 3, ch = id & 7;
            const int n = n0 + row, k = k0 + ch * EPC;
            rw[it] = (n < N && k < K) ? *reinterpret_cast<const u32x4 *>(p.W + (size_t)n * p.ldw + k) : (u32x4){0, 0, 0, 0};
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int id = it * 256 + tid, row = id >> 3, ch = id & 7;
            *reinterpret_cast<u32x4 *>(a_area + (buf * BM + row) * STRIDE + ch * 16) = ra[it];
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int id = it * 256 + tid, row = id >> 3, ch = id & 7;
            *reinterpret_cast<u32x4 *>(w_area + (buf * BN + row) * STRIDE + ch * 16) = rw[it];
        }
    };
    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nk = (K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile((kt + 1) * BK);
        const unsigned char *sa = a_area + (buf * BM + wm * (BM / 2) + r16) * STRIDE;
        const unsigned char *sw = w_area + (buf * BN + wn * (BN / 2) + r16) * STRIDE;
#pragma unroll
        for (int kc = 0; kc < BK / 32; ++kc) {
            if (kt * BK + kc * 32 < Kp) {
                const int off = kc * 32 * (int)sizeof(T) + g * 8 * (int)sizeof(T);
                typename FragOf<T>::type a[MI], b[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i] = load_frag(reinterpret_cast<const T *>(sa + i * 16 * STRIDE + off));
#pragma unroll
                for (int j = 0; j < NI; ++j) b[j] = load_frag(reinterpret_cast<const T *>(sw + j * 16 * STRIDE + off));
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = mma16(b[j], a[i], acc[i][j]);
            }
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }
    gemm_epilogue<BM, BN, MI, NI, Epi>(acc, smem, m0, n0, M, N, epi);
}

// ---- host side ---------------------------------------------------------------------------------------
static inline hipError_t raise_lds_limit(const void *kern, size_t lds) {
    if (lds <= 64 * 1024) return hipSuccess;
    static std::set<const void *> raised;      // per kernel instantiation, once
    if (raised.count(kern)) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) raised.insert(kern);
    return e;
}

template <typename T, int BM, int BN, int NST, typename Epi>
static inline hipError_t launch_ring_cfg(hipStream_t s, const GemmArgs<T> &a, const Epi &epi) {
    const size_t lds = std::max((size_t)NST * (BM + BN) * 128, epi_lds_bytes<Epi, BM, BN>());
    auto kern = gemm_ring_kernel<T, BM, BN, NST, Epi>;
    hipError_t e = raise_lds_limit((const void *)kern, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(ceil_div(a.N, BN) * ceil_div(a.M, BM)), dim3(256), lds, s, a, epi);
    return hipGetLastError();
}
template <typename T, int BM, int BN, typename Epi>
static inline hipError_t launch_stream_cfg(hipStream_t s, const GemmArgs<T> &a, const Epi &epi) {
    const size_t lds = std::max((size_t)2 * (BM + BN) * 144, epi_lds_bytes<Epi, BM, BN>());
    auto kern = gemm_stream_kernel<T, BM, BN, Epi>;
    hipError_t e = raise_lds_limit((const void *)kern, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(ceil_div(a.N, BN) * ceil_div(a.M, BM)), dim3(256), lds, s, a, epi);
    return hipGetLastError();
}

// Requirements (checked at model build): K % (16/sizeof(T)) == 0, lda/ldw multiples of 16 bytes, 16-byte aligned bases.
template <typename T, typename Epi>
static inline hipError_t launch_gemm(hipStream_t s, const T *A, int lda, const T *W, int ldw, int M, int N, int K, const Epi &epi) {
    GemmArgs<T> a{A, lda, W, ldw, M, N, K, 0};
    constexpr int BK = 128 / (int)sizeof(T);
    const long t128 = (long)ceil_div(M, 128) * ceil_div(N, 128);
    // the ring kernel addresses its operand rows with 32-bit byte offsets from a uniform base
    const bool off32 = (size_t)M * lda * sizeof(T) < ((size_t)1 << 32) - 256 && (size_t)N * ldw * sizeof(T) < ((size_t)1 << 32) - 256;
    if (K % BK == 0 && off32) {
        if (t128 >= 1024) return launch_ring_cfg<T, 128, 128, 2, Epi>(s, a, epi);
        if (N >= 512) return launch_ring_cfg<T, 64, 128, 3, Epi>(s, a, epi);
        return launch_ring_cfg<T, 64, 64, 3, Epi>(s, a, epi);
    }
    if (t128 >= 1024) return launch_stream_cfg<T, 128, 128, Epi>(s, a, epi);
    return launch_stream_cfg<T, 64, 64, Epi>(s, a, epi);
}

// ---- row-complete ring kernel: 48 rows x 256 columns per workgroup ------------------------------------------------
// For the N == encoder_dim <= 256 products with the residual + LayerNorm epilogue.  M = 9600 rows over 256 CUs is
// 37.5 rows per CU: 48-row workgroups give 200 workgroups = ONE wave of workgroups at one per CU (64-row ones leave 106
// CUs idle, 32-row ones need two rounds).  4 waves side by side (1 x 4): each wave owns all 48 rows x 64 columns
// (3 x 4 MFMA tiles: 12 MFMAs per 7 fragment reads).  The A tile is DMA'd as 64 rows (rows >= 48 are never read), the
// ring is 3 deep (one workgroup per CU: the ring is the latency hiding).
template <typename T, typename Epi>
__global__ __launch_bounds__(256) void gemm_rowln48_kernel(GemmArgs<T> p, Epi epi) {
    constexpr int BMC = 48, BMD = 64, BN = 256, NST = 3;
    constexpr int ROWB = 128, BK = ROWB / sizeof(T), EPC = 16 / sizeof(T);
    constexpr int MI = 3, NI = 4;
    constexpr int PER = BMD / 32 + BN / 32;
    constexpr int STAGE = (BMD + BN) * ROWB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int M = p.M, N = p.N, K = p.K;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4, swz = r16 & 7;
    const int m0 = blockIdx.x * BMC;
    const int nk = K / BK;
    auto issue = [&](int kt) {
        unsigned char *st = smem + (kt % NST) * STAGE;
        const int k0 = kt * BK, lrow = lane >> 3, cpos = lane & 7;
#pragma unroll
        for (int i = 0; i < BMD / 32; ++i) {
            const int rg = wave + 4 * i, row = rg * 8 + lrow;
            const T *src = p.A + (size_t)min(m0 + row, M - 1) * p.lda + k0 + ((cpos ^ (row & 7)) * EPC);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(st + rg * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) {
            const int rg = wave + 4 * i, row = rg * 8 + lrow;
            const T *src = p.W + (size_t)min(row, N - 1) * p.ldw + k0 + ((cpos ^ (row & 7)) * EPC);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(st + BMD * ROWB + rg * 1024), 16, 0, 0);
        }
    };
    // residual rows of this wave's share of the LayerNorm epilogue: requested first, consumed last
    typename Epi::Rows4 xr[BMC / 16];
#pragma unroll
    for (int it = 0; it < BMC / 16; ++it) xr[it] = epi.rows4_load(m0 + wave * 4 + 16 * it, min(M, m0 + BMC), lane);
    issue(0);
    if (nk > 1) issue(1);
    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) wait_vmcnt<PER>(); else wait_vmcnt<0>();      // at this point tiles 0 .. kt+1 have been issued
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) issue(kt + 2);
        const unsigned char *st = smem + (kt % NST) * STAGE;
        const unsigned char *sa = st + r16 * ROWB;
        const unsigned char *sw = st + BMD * ROWB + (wave * 64 + r16) * ROWB;
#pragma unroll
        for (int kc = 0; kc < BK / 32; ++kc) {
            typename FragOf<T>::type a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = lds_frag_swz(sa + i * 16 * ROWB, kc, g, swz, T());
#pragma unroll
            for (int j = 0; j < NI; ++j) b[j] = lds_frag_swz(sw + j * 16 * ROWB, kc, g, swz, T());
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = mma16(b[j], a[i], acc[i][j]);
        }
    }
    __syncthreads();
    constexpr int RS = BN * 4 + 16;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int row = 16 * i + r16, n = wave * 64 + 16 * j + 4 * g;
            const float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            float r[4];
            epi.transform(n, v, r);
            *reinterpret_cast<f32x4 *>(smem + row * RS + n * 4) = (f32x4){r[0], r[1], r[2], r[3]};
        }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < BMC / 16; ++it) {
        const int rr = wave * 4 + 16 * it;
        epi.rows4(m0 + rr, min(M, m0 + BMC), reinterpret_cast<const float *>(smem + rr * RS), RS / 4, lane, xr[it]);
    }
}

// Row-complete product (N == encoder_dim <= 256) with the fused residual + LayerNorm epilogue.
template <typename T> static inline bool gemm_rowln_supported(int N) { return N <= 256 && (N & 3) == 0; }
template <typename T, typename Epi>
static inline hipError_t launch_gemm_rowln(hipStream_t s, const T *A, int lda, const T *W, int ldw, int M, int N, int K, const Epi &epi) {
    GemmArgs<T> a{A, lda, W, ldw, M, N, K, 0};
    constexpr int BK = 128 / (int)sizeof(T);
    if (K % BK == 0) {
        const size_t lds = (size_t)3 * (64 + 256) * 128;          // >= the 48 x (256*4+16) staged tile
        auto kern = gemm_rowln48_kernel<T, Epi>;
        hipError_t e = raise_lds_limit((const void *)kern, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(ceil_div(M, 48)), dim3(256), lds, s, a, epi);
        return hipGetLastError();
    }
    return launch_stream_cfg<T, 32, 256, Epi>(s, a, epi);
}

// Split-K product for the frontend output linear (M x 256 x 6144): `splits` partial sums [z][M][N] fp32 (no bias), summed by
// splitk_reduce_ln_kernel.  One launch, grid.y = splits; 128 x 128 tiles.
#ifndef COCR_FO_BM
#define COCR_FO_BM 128
#define COCR_FO_BN 128
#define COCR_FO_NST 2
#endif
// Batched exact-fp32 products (GemmArgs::z_div): out_z (M x N, row stride ldo) = A_z (M x K) W_z (N x K)^T, K % 32 == 0, 64 x 64 tiles.
static inline hipError_t launch_gemm_batched_f32(hipStream_t s, GemmArgs<float> a, float *out, int ldo, int batches) {
    constexpr int BM = 64, BN = 64, NST = 3;
    if (a.K % 32 || a.z_div < 1 || batches < 1) return hipErrorInvalidValue;
    EpiStoreF32 e{out, ldo, nullptr, a.N};
    const size_t lds = std::max((size_t)NST * (BM + BN) * 128, epi_lds_bytes<EpiStoreF32, BM, BN>());
    auto kern = gemm_ring_kernel<float, BM, BN, NST, EpiStoreF32>;
    hipError_t err = raise_lds_limit((const void *)kern, lds);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(kern, dim3(ceil_div(a.N, BN) * ceil_div(a.M, BM), batches), dim3(256), lds, s, a, e);
    return hipGetLastError();
}

template <typename T>
static inline hipError_t launch_gemm_splitk(hipStream_t s, const T *A, int lda, const T *W, int ldw, int M, int N, int K, int splits, float *partial) {
    constexpr int BK = 128 / (int)sizeof(T);
    constexpr int BM = COCR_FO_BM, BN = COCR_FO_BN, NST = COCR_FO_NST;
    if (K % (splits * BK)) return hipErrorInvalidValue;
    if ((size_t)M * lda * sizeof(T) >= ((size_t)1 << 32) - 256 || (size_t)N * ldw * sizeof(T) >= ((size_t)1 << 32) - 256) return hipErrorInvalidValue;
    GemmArgs<T> a{A, lda, W, ldw, M, N, K / splits, K / splits};
    EpiStoreF32 e{partial, N, nullptr, N};
    e.zstride = (size_t)M * N;
    const size_t lds = std::max((size_t)NST * (BM + BN) * 128, epi_lds_bytes<EpiStoreF32, BM, BN>());
    auto kern = gemm_ring_kernel<T, BM, BN, NST, EpiStoreF32>;
    hipError_t err = raise_lds_limit((const void *)kern, lds);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(kern, dim3(ceil_div(N, BN) * ceil_div(M, BM), splits), dim3(256), lds, s, a, e);
    return hipGetLastError();
}

// out (M, N) (or `splits` partial sums at partial + z M N, to be added by the caller) = A^T B for K-major A (K, M) and B (K, N)
static inline hipError_t launch_gemm_tn(hipStream_t s, const bf16_t *A, int lda, const bf16_t *B, int ldb, int M, int N, int K, int splits, float *out) {
    constexpr int BM = 128, BN = 128, NST = 4;
    if (splits < 1 || K % (splits * 32) || M % 8 || N % 8 || lda % 8 || ldb % 8) return hipErrorInvalidValue;
    if ((size_t)K * lda * 2 >= ((size_t)1 << 32) - 256 || (size_t)K * ldb * 2 >= ((size_t)1 << 32) - 256) return hipErrorInvalidValue;      // (32-bit row offsets inside a split)
    GemmArgs<bf16_t> a{A, lda, B, ldb, M, N, K / splits, splits > 1 ? K / splits : 0};
    EpiStoreF32 e{out, N, nullptr, N};
    e.zstride = (size_t)M * N;
    const size_t lds = std::max((size_t)NST * 32 * (BM + BN) * 2, epi_lds_bytes<EpiStoreF32, BM, BN>());
    auto kern = gemm_tn_kernel<BM, BN, NST>;
    hipError_t err = raise_lds_limit((const void *)kern, lds);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(kern, dim3(ceil_div(N, BN) * ceil_div(M, BM), splits), dim3(256), lds, s, a, e);
    return hipGetLastError();
}

