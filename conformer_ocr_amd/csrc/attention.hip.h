// Relative-position multi-head self-attention core (reference conformer/attention.py:87-101 with
// _relative_shift :105-113 and the positional slice of embedding.py:66), flash style:
//
//   score[b,h,i,j] = ((q_i + u_h) . k_j  +  (q_i + v_h) . P_h[CEN - (i - j)]) / sqrt(d_head)
//   ctx[b,i,h,:]   = sum_j softmax_j(score)[j] v_j          over ALL j in [0,T): no mask
//
// P is the per-layer table PE Wpos^T precomputed for all 9999 relative positions (row CEN = 4999 is
// relative position 0), so the reference's pad/view "relative shift" is just the index CEN - (i - j).
//
// Decomposition: workgroup = one (line, head, 64-query tile); each of its 4 waves owns 16 queries and
// walks the keys in tiles of 32 with an online softmax.  All products are computed TRANSPOSED (keys /
// positions / head-dim on the MFMA row side, the wave's 16 queries on the column side), so that a
// query's scores live in one lane column: the softmax statistics are per-lane scalars (plus two
// cross-quad shuffles) and the exponentiated scores are already the B operand of the P.V product.
//   S^T  (32 keys x 16 q)   = K_tile      . (Q+u)^T     2 row tiles x dhp/32 k-chunks
//   R^T  (48 rows x 16 q)   = P_band      . (Q+v)^T     band rows rbase .. rbase+46, rbase = CEN-(i0+15)+j0
//   pos^T[jl][il]           = R^T[15 - il + jl][il]      the "shift": through a per-wave LDS tile
//   O^T  (dhp x 16 q)      += V^T_tile    . P^T          k-chunk = the 32 keys of the tile
// Operands come straight from global memory (K, V^T and the band of P are L2 resident: one (b,h) is
// 2 x 38 KB at T=300); layouts q,k [B][h][Tp][dhp], vt [B][h][dhp][Tp], P [9999][h][dhp].
#pragma once
#include "common.hip.h"

#define COCR_POS_CENTER 4999
#define COCR_POS_ROWS 9999

template <typename T, int DHP>   // DHP = padded head dim, multiple of 32
__global__ __launch_bounds__(256) void relpos_attention_kernel(const T *__restrict__ q, const T *__restrict__ k, const T *__restrict__ vt,
                                                               const T *__restrict__ ptab, const float *__restrict__ ub,
                                                               const float *__restrict__ vb, T *__restrict__ ctx,
                                                               int Tn, int Tp, int heads, int dh, float scale) {
    constexpr int KC = DHP / 32;     // k-chunks over the head dim
    constexpr int DT = DHP / 16;     // 16-row tiles of O^T
    constexpr int SK = 20;           // LDS row stride (floats) of the shift tile: conflict-free write and skewed read
    __shared__ float skew[4][48 * SK];
    typedef typename FragOf<T>::type frag_t;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int il = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y, b = bh / heads, hh = bh - b * heads;
    const int i0 = blockIdx.x * 64 + wave * 16;
    // queries beyond T only exist to keep the wave uniform: they read row T-1 and store nothing
    const int iq = min(i0 + il, Tn - 1);

    const T *qrow = q + ((size_t)bh * Tp + iq) * DHP;
    const T *kbase = k + (size_t)bh * Tp * DHP;
    const T *vbase = vt + (size_t)bh * DHP * Tp;
    const int prow = heads * DHP;                      // elements per table row
    const T *pbase = ptab + hh * DHP;

    // B operands: (q + u) and (q + v) of this lane's query, per k-chunk; padded dims stay zero
    frag_t qu[KC], qv[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        const frag_t qq = load_frag(qrow + c * 32 + 8 * g);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = c * 32 + 8 * g + j;
            const float x = to_f32(qq[j]);
            qu[c][j] = d < dh ? from_f32<T>(x + ub[hh * dh + d]) : (T)0.0f;
            qv[c][j] = d < dh ? from_f32<T>(x + vb[hh * dh + d]) : (T)0.0f;
        }
    }

    f32x4 o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    float *sk = skew[wave];

    for (int j0 = 0; j0 < Tn; j0 += 32) {
        // ---- content scores, transposed: rows = keys j0 + 16 tt + (4g + reg), column = query il
        f32x4 sc[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            sc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int jr = min(j0 + 16 * tt + il, Tn - 1);        // A-operand row of this lane (clamped; masked below)
#pragma unroll
            for (int c = 0; c < KC; ++c) sc[tt] = mma16(load_frag(kbase + (size_t)jr * DHP + c * 32 + 8 * g), qu[c], sc[tt]);
        }
        // ---- positional scores on the band of P this (query tile, key tile) pair touches
        const int rbase = COCR_POS_CENTER - (i0 + 15) + j0;
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) {
            f32x4 rp = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int pr = min(max(rbase + 16 * mt + il, 0), COCR_POS_ROWS - 1);
#pragma unroll
            for (int c = 0; c < KC; ++c) rp = mma16(load_frag(pbase + (size_t)pr * prow + c * 32 + 8 * g), qv[c], rp);
#pragma unroll
            for (int r = 0; r < 4; ++r) sk[(16 * mt + 4 * g + r) * SK + il] = rp[r];
        }
        __syncthreads();
        // ---- shift, scale, mask padded keys of the last tile
        float tmax = -INFINITY;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int jl = 16 * tt + 4 * g + r;
                float s = (sc[tt][r] + sk[(15 - il + jl) * SK + il]) * scale;
                s = (j0 + jl < Tn) ? s : -INFINITY;
                sc[tt][r] = s;
                tmax = fmaxf(tmax, s);
            }
        __syncthreads();
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);                    // finite: key j0 is always valid
        const float alpha = __expf(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
        frag_t pb;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(sc[tt][r] - m_new);
                psum += p;
                pb[4 * tt + r] = from_f32<T>(p);
            }
        l_run = l_run * alpha + psum;
        // ---- O^T = alpha O^T + V^T_tile . P^T ; k order inside the chunk: element 4tt + r of quad g <-> key 16 tt + 4g + r
#pragma unroll
        for (int d = 0; d < DT; ++d) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[d][r] *= alpha;
            const T *vr = vbase + (size_t)(16 * d + il) * Tp + j0 + 4 * g;
            frag_t va;
            if constexpr (sizeof(T) == 2) {
                const bf16x4 lo = *reinterpret_cast<const bf16x4 *>(vr), hi = *reinterpret_cast<const bf16x4 *>(vr + 16);
#pragma unroll
                for (int r = 0; r < 4; ++r) { va[r] = lo[r]; va[4 + r] = hi[r]; }
            } else {
                const f32x4 lo = *reinterpret_cast<const f32x4 *>(vr), hi = *reinterpret_cast<const f32x4 *>(vr + 16);
#pragma unroll
                for (int r = 0; r < 4; ++r) { va[r] = lo[r]; va[4 + r] = hi[r]; }
            }
            o[d] = mma16(va, pb, o[d]);
        }
    }
    // ---- normalise and store: lane holds head dims 16 d + 4g + r of query i0 + il
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_run;
    if (i0 + il < Tn) {
        T *dst = ctx + ((size_t)b * Tn + i0 + il) * (heads * dh) + hh * dh;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int dd = 16 * d + 4 * g + r;
                if (dd < dh) dst[dd] = from_f32<T>(o[d][r] * inv);
            }
    }
}
