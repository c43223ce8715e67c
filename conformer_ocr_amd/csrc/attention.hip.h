// Relative-position multi-head self-attention core (reference conformer/attention.py:87-101 with
// _relative_shift :105-113 and the positional slice of embedding.py:66), flash style:
//
//   score[b,h,i,j] = ((q_i + u_h) . k_j  +  (q_i + v_h) . P_h[CEN - (i - j)]) / sqrt(d_head)      (`scale` = log2(e) / sqrt(d_head):
//                                                                                                  scores in log2 units, 2^x softmax)
//   ctx[b,i,h,:]   = sum_j softmax_j(score)[j] v_j          over ALL j in [0,T): no mask
//
// P is the per-layer table PE Wpos^T precomputed for all 9999 relative positions (row CEN = 4999 is
// relative position 0), so the reference's pad/view "relative shift" is just the index CEN - (i - j).
//
// Decomposition: workgroup = one (line, head, 64-query tile); each of its 4 waves owns 16 queries.  The
// keys are walked in tiles of 64: K tile, V tile and the 128-row band of P that the (64 queries x 64 keys)
// pair touches are staged in LDS once per tile (global -> registers while the previous tile is being
// computed -> LDS), and every wave reads its fragments from there.  All products are computed TRANSPOSED
// (keys / positions / head-dim on the MFMA row side, the wave's 16 queries on the column side): a query's
// scores live in one lane column, so the online-softmax statistics are per-lane scalars (plus two
// cross-quad shuffles) and the exponentiated scores are directly the B operand of the P.V product:
//   S^T  (32 keys x 16 q)   = K_sub       . (Q+u)^T     2 row tiles x dhp/32 k-chunks
//   R^T  (48 rows x 16 q)   = P_band      . (Q+v)^T     band rows of this (wave, 32-key sub-tile)
//   pos^T[jl][il]           = R^T[15 - il + jl][il]      the "shift": through a per-wave LDS tile
//   O^T  (dhp x 16 q)      += V^T_sub     . P^T          V^T fragments by ds_read_b64_tr_b16 (bf16) from the
//                                                        row-major V tile: no transposed copy of V anywhere
// Layouts: q, k, v [B][h][Tp][dhp] (head dim zero-padded to a multiple of 32), P [9999][h][dhp].
#pragma once
#include "common.hip.h"

// hardware workgroup id -> logical id such that each XCD (id % 8) owns a contiguous range of logical ids
__device__ __forceinline__ int xcd_remap_i(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// max of three without the NaN-quieting v_max x, x the compiler puts in front of every fmaxf operand (scores are never NaN: finite
// products, or -inf from the tail mask)
__device__ __forceinline__ float max3_f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// EVERY inline-assembly consumer of matrix-instruction results goes through one of these fences: the compiler's hazard
// recogniser does not see that such a statement reads registers a v_mfma has just written, and the hardware does not interlock a vector
// read on a matrix result.  The accumulators are tied to the statement ("+v": every product is issued before it, every consumer after it)
// and it holds the wait states of the longest pass of a 16x16x32 instruction (20).  Without it the resident kernel's results varied in the
// last bit from run to run (a stale score moves only the running reference); tests/test_hip_determinism.py is the guard.
#define COCR_MFMA_WAIT "s_nop 7\n\ts_nop 7\n\ts_nop 3"
__device__ __forceinline__ void mfma_results_fence(f32x4 (&a)[4]) { asm volatile(COCR_MFMA_WAIT : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])); }
__device__ __forceinline__ void mfma_results_fence(f32x4 (&a)[4], f32x4 (&b)[4]) {
    asm volatile(COCR_MFMA_WAIT : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
}
__device__ __forceinline__ void mfma_results_fence(f32x4 (&a)[4], f32x4 (&b)[4], f32x4 (&c)[4]) {
    asm volatile(COCR_MFMA_WAIT : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]),
                 "+v"(c[3]));
}

#ifndef COCR_ATT_EXP
#define COCR_ATT_EXP 0          // dev: timing experiments (wrong results): 1 no shift through LDS, 2 no positional products, 4 no exponentials,
                               // 8 no P.V products, 16 tiles staged once, 32 no content products, 64 no barriers
#endif
#define COCR_POS_MAXLEN 5000        // the reference's RelPositionalEncoding(max_len): 2 * max_len - 1 table rows, row max_len - 1 <-> relative position 0

// V^T fragment of one k-chunk (32 keys) for head-dim row tile d: element e of quad g <-> key 16 (e>>2) + 4g + (e&3)
// (the same permutation the exponentiated scores have in their accumulator registers).
template <bool SWZ>
__device__ __forceinline__ bf16x8 load_vt_frag(const unsigned char *vtile, int stride, int key0, int d, int il, int g, bf16_t) {
    // ds_read_b64_tr_b16: lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3 of a 4x16 block and
    // receives column (lane & 15) of the 4 rows.  Block rows = keys key0 + 4g + q, block columns = dims 16 d ..+15.
    // SWZ: 128-byte rows, 16-byte chunk c of row r stored at chunk c ^ (r & 7) (key0 is a multiple of 8; the second read is 16 rows on)
    const int row = key0 + 4 * g + (il >> 2);
    const unsigned char *p0 = SWZ ? vtile + row * stride + (((2 * d + ((il & 3) >> 1)) ^ (row & 7)) << 4) + 8 * (il & 1)
                                  : vtile + row * stride + (16 * d + 4 * (il & 3)) * 2;
    typedef __attribute__((address_space(3))) bf16x4 *lp;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)p0);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(p0 + 16 * stride));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
template <bool SWZ>
__device__ __forceinline__ f32x8 load_vt_frag(const unsigned char *vtile, int stride, int key0, int d, int il, int g, float) {
    static_assert(!SWZ, "the swizzled image is the bf16 / 128-byte-row form");
    f32x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e)
        r[e] = *reinterpret_cast<const float *>(vtile + (key0 + 16 * (e >> 2) + 4 * g + (e & 3)) * stride + (16 * d + il) * 4);
    return r;
}

template <typename T, int DHP>   // DHP = padded head dim, multiple of 32
__global__ __launch_bounds__(256) void relpos_attention_kernel(const T *__restrict__ q, const T *__restrict__ k, const T *__restrict__ v,
                                                               const T *__restrict__ ptab, const float *__restrict__ ub,
                                                               const float *__restrict__ vb, T *__restrict__ ctx,
                                                               int Tn, int Tp, int heads, int dh, float scale, int pos_center, unsigned long long *stamps) {
    constexpr int KC = DHP / 32;                      // k-chunks over the head dim
    constexpr bool MERGE = sizeof(T) == 2 && DHP <= 64;   // one softmax step per 64 keys (bf16: the registers allow it at 3 waves per SIMD)
    constexpr int DT = DHP / 16;                      // 16-row tiles of O^T
    constexpr int SK = 20;                            // row stride (floats) of the shift tile: conflict-free write and skewed read
    constexpr int RB = DHP * (int)sizeof(T);          // bytes per operand row
    // LDS rows: 128-byte rows (bf16, d_head 64: the measured configuration) are stored unpadded with the 16-byte chunk index
    // XOR-ed with (row & 7): ds_read_b128 fragment reads, the ds_read_b64_tr_b16 reads of V and the staging ds_write_b128 are all
    // bank-conflict free (with the former 144-byte padded rows 7 of 8 fragment reads were 2-way conflicts); other row sizes keep
    // a 16-byte pad.
    constexpr bool SWZ = RB == 128 && sizeof(T) == 2;
    constexpr int RS = SWZ ? RB : RB + 16;
    constexpr int CPR = RB / 16;                      // 16-byte chunks per row
    constexpr int KV_IT = 64 * CPR / 256, P_IT = 128 * CPR / 256;
    typedef typename FragOf<T>::type frag_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char att_smem[];
    unsigned char *ks = att_smem, *vs = ks + 64 * RS, *ps = vs + 64 * RS;
    float *skew = reinterpret_cast<float *>(ps + 128 * RS);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int il = lane & 15, g = lane >> 4;
    // byte offset of this lane's 16 (bf16) / 32 (fp32) bytes of k-chunk c inside a tile row whose index is il mod 8 (every
    // fragment row base below is a multiple of 16 rows)
    auto frag_off = [&](int c) { return SWZ ? (((4 * c + g) ^ (il & 7)) << 4) : (c * 32 + 8 * g) * (int)sizeof(T); };
    // XCD-aware order: the query tiles of one (line, head) re-read the same K / V / P rows; consecutive LOGICAL workgroups share
    // an XCD (hardware id % 8 picks the XCD), so those rows are fetched into one L2 instead of up to five
    // (PMC: 61 MB of HBM traffic per launch against 21 MB algorithmic with the plain (x, y) order).
    const int nqt = gridDim.x, logical = xcd_remap_i(blockIdx.x + nqt * blockIdx.y, nqt * gridDim.y);
    const int qtile = logical % nqt;
    const int bh = logical / nqt, b = bh / heads, hh = bh - b * heads;
    const int i0b = qtile * 64, i0 = i0b + wave * 16;
    const int iq = min(i0 + il, Tn - 1);              // queries beyond T read row T-1 and store nothing

#ifdef COCR_CHAIN_STAMPS_BUILD
    int nstamp = 0;
    auto stamp = [&]() {
        if (stamps && logical == 26 && tid == 0) stamps[nstamp] = __builtin_readcyclecounter();
        ++nstamp;
    };
#else
    auto stamp = [&]() {};
#endif
    stamp();
#ifdef COCR_CHAIN_STAMPS_BUILD
    if (stamps && tid == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        stamps[64 + 3 * logical] = wall_clock64();
        stamps[64 + 3 * logical + 2] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
    const T *kbase = k + (size_t)bh * Tp * DHP;
    const T *vbase = v + (size_t)bh * Tp * DHP;
    const int prow = heads * DHP;
    const T *pbase = ptab + hh * DHP;

    // tile loads: K/V rows j0 .. j0+63, band rows B0 .. B0+127, B0 = CEN - (i0b + 63) + j0.  No clamps: q / k / v hold Tp rows per
    // (line, head) with Tp a multiple of 64 (rows beyond T are zero and masked in the last tile), and the band stays inside the table
    // (cocr_forward rebuilds the tables for a line longer than they cover).  Each lane's byte offsets inside a tile are constants; the tile bases are
    // uniform, so a tile costs its 8 load instructions and two scalar adds (the clamped per-row addresses were ~40 vector
    // instructions per tile in a loop whose SIMDs are issue-bound).
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 rk[KV_IT], rv[KV_IT], rp[P_IT];
    unsigned kvoff[KV_IT], poff[P_IT];
#pragma unroll
    for (int it = 0; it < KV_IT; ++it) { const int id = it * 256 + tid, row = id / CPR, ch = id - row * CPR; kvoff[it] = (unsigned)row * RB + ch * 16; }
#pragma unroll
    for (int it = 0; it < P_IT; ++it) { const int id = it * 256 + tid, row = id / CPR, ch = id - row * CPR; poff[it] = (unsigned)row * (unsigned)prow * (unsigned)sizeof(T) + ch * 16; }
    // The band of consecutive key tiles overlaps by half (it moves by 64 rows per tile): the 128 band rows in LDS are a ring, local band row
    // lr of tile t sits in slot (lr + 64 (t & 1)) & 127, and every tile after the first stages only its 64 new rows (24 KB instead of 32 KB
    // per tile through the memory pipeline: staging is a quarter of this kernel's time, timing experiment COCR_ATT_EXP=16).
    auto load_tile = [&](int j0, bool first) {
        const unsigned char *kt = reinterpret_cast<const unsigned char *>(kbase) + (size_t)j0 * RB;
        const unsigned char *vt = reinterpret_cast<const unsigned char *>(vbase) + (size_t)j0 * RB;
        const unsigned char *pt = reinterpret_cast<const unsigned char *>(pbase) + (size_t)(pos_center - (i0b + 63) + j0) * prow * sizeof(T);
#pragma unroll
        for (int it = 0; it < KV_IT; ++it) {
            rk[it] = *reinterpret_cast<const u32x4 *>(kt + kvoff[it]);
            rv[it] = *reinterpret_cast<const u32x4 *>(vt + kvoff[it]);
        }
#pragma unroll
        for (int it = 0; it < P_IT; ++it) if (first || it >= P_IT / 2) rp[it] = *reinterpret_cast<const u32x4 *>(pt + poff[it]);
    };
    auto store_tile = [&](int t) {
#pragma unroll
        for (int it = 0; it < KV_IT; ++it) {
            const int id = it * 256 + tid, row = id / CPR, ch = id - row * CPR;
            const int cs = SWZ ? (ch ^ (row & 7)) : ch;
            *reinterpret_cast<u32x4 *>(ks + row * RS + cs * 16) = rk[it];
            *reinterpret_cast<u32x4 *>(vs + row * RS + cs * 16) = rv[it];
        }
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            if (t > 0 && it < P_IT / 2) continue;           // (uniform) rows 0..63 of this tile's band are rows 64..127 of the previous one's
            const int id = it * 256 + tid, lr = id / CPR, ch = id - lr * CPR, row = (lr + 64 * (t & 1)) & 127;
            *reinterpret_cast<u32x4 *>(ps + row * RS + (SWZ ? (ch ^ (row & 7)) : ch) * 16) = rp[it];
        }
    };
    load_tile(0, true);

    // B operands: (q + u) and (q + v) of this lane's query, per k-chunk; padded dims stay zero.  All loads (q fragments
    // and the 8 consecutive bias values of each) are issued before the first use: one round trip, not one per element.
    frag_t qu[KC], qv[KC];
    if (dh == DHP) {
        // common case (d_head a multiple of 32): no padded dims -- 16-byte loads only, no per-element selects
        const T *qrow = q + ((size_t)bh * Tp + iq) * DHP;
        frag_t qq[KC];
        f32x4 u4[KC][2], v4[KC][2];
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            qq[c] = load_frag(qrow + c * 32 + 8 * g);
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                u4[c][hlf] = *reinterpret_cast<const f32x4 *>(ub + hh * DHP + c * 32 + 8 * g + 4 * hlf);
                v4[c][hlf] = *reinterpret_cast<const f32x4 *>(vb + hh * DHP + c * 32 + 8 * g + 4 * hlf);
            }
        }
#pragma unroll
        for (int c = 0; c < KC; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = to_f32(qq[c][j]);
                qu[c][j] = from_f32<T>((x + u4[c][j >> 2][j & 3]) * scale);      // 1/sqrt(d_head) folded into the query operands
                qv[c][j] = from_f32<T>((x + v4[c][j >> 2][j & 3]) * scale);
            }
    } else {
        const T *qrow = q + ((size_t)bh * Tp + iq) * DHP;
        frag_t qq[KC];
        f32x4 u4[KC][2], v4[KC][2];
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            qq[c] = load_frag(qrow + c * 32 + 8 * g);
            const int d0 = min(c * 32 + 8 * g, max(dh - 8, 0));     // clamped: values beyond dh are masked below
            if ((dh & 7) == 0) {
#pragma unroll
                for (int hlf = 0; hlf < 2; ++hlf) {
                    u4[c][hlf] = *reinterpret_cast<const f32x4 *>(ub + hh * dh + d0 + 4 * hlf);
                    v4[c][hlf] = *reinterpret_cast<const f32x4 *>(vb + hh * dh + d0 + 4 * hlf);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int d = min(c * 32 + 8 * g + j, dh - 1);
                    u4[c][j >> 2][j & 3] = ub[hh * dh + d];
                    v4[c][j >> 2][j & 3] = vb[hh * dh + d];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < KC; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = c * 32 + 8 * g + j;
                const float x = to_f32(qq[c][j]);
                qu[c][j] = d < dh ? from_f32<T>((x + u4[c][j >> 2][j & 3]) * scale) : (T)0.0f;      // 1/sqrt(d_head) folded into the query operands
                qv[c][j] = d < dh ? from_f32<T>((x + v4[c][j >> 2][j & 3]) * scale) : (T)0.0f;
            }
    }

    f32x4 o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 negm4 = {0.f, 0.f, 0.f, 0.f};               // -m_run in the accumulator layout (64-key form); zero until the first tile has set it
    f32x4 osum = {0.f, 0.f, 0.f, 0.f};                // 64-key form: row 0 (register 0 of lanes 0..15) = softmax denominator of query `lane`
    frag_t ones;                                      // A operand whose row 0 is all ones
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = il == 0 ? (T)1.0f : (T)0.0f;
    float *sk = skew + wave * 48 * SK;

    stamp();
    for (int j0 = 0; j0 < Tn; j0 += 64) {
        if constexpr (!(COCR_ATT_EXP & 64)) __syncthreads();                       // every wave is done with the previous tile
        stamp();
        const int bslot = 64 * ((j0 >> 6) & 1);    // ring offset of this tile's band rows
        if (!(COCR_ATT_EXP & 16) || j0 == 0) store_tile(j0 >> 6);
        if constexpr (!(COCR_ATT_EXP & 64)) __syncthreads();
        stamp();
        if (!(COCR_ATT_EXP & 16)) load_tile(j0 + 64 < Tn ? j0 + 64 : j0, false);   // in flight during the compute below (unconditional: the staging registers stay registers)
        if constexpr (MERGE) {
            // ---- one softmax step per 64-key tile: the two 32-key halves still take the shift tile in turn (it holds 48 band rows),
            // but the cross-lane maximum (two dependent ds_bpermute round trips), the rescale test, the exponentials and the P.V
            // product run once per tile on 16 scores per lane -- the per-wave dependent chain is what bounds this kernel.
            // Two folds keep the VALU out of the score path (the SIMDs of the CUs that hold three workgroups are issue-bound):
            //  * the positional products start from C = -m_run (the running reference of this lane's query), so what comes back
            //    from the shift tile is already "positional score - reference";
            //  * those shifted values are the C operand of the content products: no add per score either.
            f32x4 sc[4];
            f32x4 keep = {0.f, 0.f, 0.f, 0.f};                    // band tile 2: last of the first half, first of the second
            const int lb0 = 48 - 16 * wave;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int ml = 0; ml < 3; ++ml) {
                    f32x4 rr;
                    if (half == 1 && ml == 0) rr = keep;
                    else {
                        rr = negm4;
                        const unsigned char *pr = ps + ((lb0 + 16 * (2 * half + ml) + il + bslot) & 127) * RS;
                        if constexpr (!(COCR_ATT_EXP & 2)) {
#pragma unroll
                            for (int c = 0; c < KC; ++c) rr = mma16(load_frag(reinterpret_cast<const T *>(pr + frag_off(c))), qv[c], rr); }
                    }
                    if (half == 0 && ml == 2) keep = rr;
                    if constexpr (COCR_ATT_EXP & 1) { if (ml < 2) sc[2 * half + ml] = rr; }
                    else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sk[(16 * ml + 4 * g + r) * SK + il] = rr[r];
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the shift tile is private to the wave; its LDS operations execute in order
                if constexpr (!(COCR_ATT_EXP & 1)) {
#pragma unroll
                for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sc[2 * half + tl][r] = sk[(15 - il + 16 * tl + 4 * g + r) * SK + il];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the tile is overwritten
#pragma unroll
                for (int tl = 0; tl < 2; ++tl) {
                    const int tt = 2 * half + tl;
                    const unsigned char *kr = ks + (16 * tt + il) * RS;
                    if constexpr (!(COCR_ATT_EXP & 32)) {
#pragma unroll
                        for (int c = 0; c < KC; ++c) sc[tt] = mma16(load_frag(reinterpret_cast<const T *>(kr + frag_off(c))), qu[c], sc[tt]); }
                }
            }
            // sc = score - m_run (log2 units).  The maxima are inline assembly (max3_f): fence first
            mfma_results_fence(sc);
            float tmax = max3_f(sc[0][0], sc[0][1], sc[0][2]);
            tmax = max3_f(tmax, sc[0][3], sc[1][0]);
            tmax = max3_f(tmax, sc[1][1], sc[1][2]);
            tmax = max3_f(tmax, sc[1][3], sc[2][0]);
            tmax = max3_f(tmax, sc[2][1], sc[2][2]);
            tmax = max3_f(tmax, sc[2][3], sc[3][0]);
            tmax = max3_f(tmax, sc[3][1], sc[3][2]);
            tmax = max3_f(tmax, sc[3][3], sc[3][3]);
            if (j0 + 64 > Tn) {                                      // uniform: only the last tile has keys beyond T (a real branch, see below)
                asm volatile("; keys beyond the line" ::: "memory");
                tmax = -INFINITY;
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (j0 + 16 * tt + 4 * g + r >= Tn) sc[tt][r] = -INFINITY;
                        tmax = fmaxf(tmax, sc[tt][r]);
                    }
            }
            { const float x = __shfl_xor(tmax, 16, 64); tmax = max3_f(tmax, x, x); }
            { const float x = __shfl_xor(tmax, 32, 64); tmax = max3_f(tmax, x, x); }
            // Lazy rescaling (see the 32-key form below): the reference moves only when some query of the wave exceeds it by > LAZY;
            // the first tile sets it (key 0 is valid: finite).
            constexpr float LAZY = 8.0f;
            if (j0 == 0 || __builtin_amdgcn_ballot_w64(tmax > LAZY) != 0) {
                const float delta = j0 == 0 ? tmax : fmaxf(tmax, 0.f);
                const float alpha = j0 == 0 ? 1.0f : __builtin_amdgcn_exp2f(-delta);
                m_run = j0 == 0 ? tmax : m_run + delta;
                negm4 = (f32x4){-m_run, -m_run, -m_run, -m_run};
#pragma unroll
                for (int r = 0; r < 4; ++r) osum[r] *= alpha;
#pragma unroll
                for (int d = 0; d < DT; ++d)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[d][r] *= alpha;
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sc[tt][r] -= delta;
            }
            frag_t pb[2];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) pb[tt >> 1][4 * (tt & 1) + r] = from_f32<T>((COCR_ATT_EXP & 4) ? sc[tt][r] : __builtin_amdgcn_exp2f(sc[tt][r]));
            // the softmax denominators on the matrix cores as well: one more row tile whose row 0 is all ones (2 MFMAs per key tile
            // instead of 16 VALU adds; the sum is over the SAME bf16-rounded probabilities the numerator uses)
            osum = mma16(ones, pb[0], osum);
            osum = mma16(ones, pb[1], osum);
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                if constexpr (COCR_ATT_EXP & 8) { asm volatile("" : "+v"(o[d]) : "v"(pb[0]), "v"(pb[1])); continue; }
                o[d] = mma16(load_vt_frag<SWZ>(vs, RS, 0, d, il, g, T()), pb[0], o[d]);
                o[d] = mma16(load_vt_frag<SWZ>(vs, RS, 32, d, il, g, T()), pb[1], o[d]);
            }
            stamp();
        } else {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int js = j0 + 32 * s2;       // first key of this 32-key sub-tile
            if (js >= Tn) continue;            // uniform over the workgroup (no break: keeps the loop fully unrolled)
            // ---- content scores, transposed: rows = keys js + 16 tt + (4g + reg), column = query il
            f32x4 sc[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                sc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const unsigned char *kr = ks + (32 * s2 + 16 * tt + il) * RS;
#pragma unroll
                for (int c = 0; c < KC; ++c) sc[tt] = mma16(load_frag(reinterpret_cast<const T *>(kr + frag_off(c))), qu[c], sc[tt]);
            }
            // ---- positional scores: band rows lb + 16 mt + il of the staged band, lb = (48 - 16 wave) + 32 s2
            const int lb = 48 - 16 * wave + 32 * s2;
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
                f32x4 rr = (f32x4){0.f, 0.f, 0.f, 0.f};
                const unsigned char *pr = ps + ((lb + 16 * mt + il + bslot) & 127) * RS;
#pragma unroll
                for (int c = 0; c < KC; ++c) rr = mma16(load_frag(reinterpret_cast<const T *>(pr + frag_off(c))), qv[c], rr);
#pragma unroll
                for (int r = 0; r < 4; ++r) sk[(16 * mt + 4 * g + r) * SK + il] = rr[r];
            }
            // the shift tile is private to the wave: LDS operations of one wave execute in order
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const bool tail = js + 32 > Tn;                      // uniform: only the last sub-tile has keys beyond T to mask
            float tmax = -INFINITY;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int jl = 16 * tt + 4 * g + r;
                    const float sv = sc[tt][r] + sk[(15 - il + jl) * SK + il];
                    sc[tt][r] = sv;
                    tmax = fmaxf(tmax, sv);
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next sub-tile overwrites the shift tile
            if (tail) {
                // a real branch (the asm statement keeps the compiler from turning it into 4 selects per score in EVERY sub-tile)
                asm volatile("; keys beyond the line" ::: "memory");
                tmax = -INFINITY;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (js + 16 * tt + 4 * g + r >= Tn) sc[tt][r] = -INFINITY;
                        tmax = fmaxf(tmax, sc[tt][r]);
                    }
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            // Lazy rescaling: the running reference m_run is only raised when some query of the wave exceeds it by more than
            // LAZY (softmax is invariant to the reference; exp(score - m_run) <= e^LAZY stays well inside fp32 / bf16 range), so
            // the accumulators are not touched by the VALU in the common case.  The first sub-tile sets the reference.
            constexpr float LAZY = 8.0f;
            if (j0 == 0 && s2 == 0) {
                m_run = tmax;                                     // finite: key 0 is valid
            } else if (__builtin_amdgcn_ballot_w64(tmax > m_run + LAZY) != 0) {
                const float m_new = fmaxf(m_run, tmax);
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                m_run = m_new;
                l_run *= alpha;
#pragma unroll
                for (int d = 0; d < DT; ++d)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[d][r] *= alpha;
            }
            float psum = 0.f;
            frag_t pb;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(sc[tt][r] - m_run);       // scores are in log2 units (see `scale`)
                    psum += p;
                    pb[4 * tt + r] = from_f32<T>(p);
                }
            l_run += psum;
            // ---- O^T += V^T_sub . P^T
#pragma unroll
            for (int d = 0; d < DT; ++d) o[d] = mma16(load_vt_frag<SWZ>(vs, RS, 32 * s2, d, il, g, T()), pb, o[d]);
            stamp();
        }
        }
    }
    stamp();
#ifdef COCR_CHAIN_STAMPS_BUILD
    if (stamps && tid == 0) stamps[64 + 3 * logical + 1] = wall_clock64();
#endif
    // ---- normalise and store: lane holds head dims 16 d + 4g + r of query i0 + il
    if constexpr (MERGE) {
        l_run = __shfl(osum[0], il, 64);              // the denominator of query il sits in lane il
    } else {
        l_run += __shfl_xor(l_run, 16, 64);
        l_run += __shfl_xor(l_run, 32, 64);
    }
    const float inv = 1.0f / l_run;
    if constexpr (sizeof(T) == 2 && DHP == 64) {
        if (dh == DHP) {
            // the measured case (bf16, d_head 64): the wave's 16 x 128 bytes go through its shift tile (free by now; 128-byte rows, 16-byte
            // chunks XOR-ed with row & 7) and leave as two instructions of eight whole 128-byte rows each.  Straight from the accumulators
            // an instruction stores 16 rows x 32 bytes: four instructions that each touch all sixteen cache lines.
            unsigned char *stg = reinterpret_cast<unsigned char *>(sk);
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const f32x4 r4 = o[d] * inv;
                const bf16x4 w = {(T)r4[0], (T)r4[1], (T)r4[2], (T)r4[3]};
                *reinterpret_cast<bf16x4 *>(stg + il * 128 + (((2 * d + (g >> 1)) ^ (il & 7)) << 4) + 8 * (g & 1)) = w;
            }
            // (LDS operations of one wave execute in order: the reads below see the writes above)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = 8 * j + (lane >> 3), ch = lane & 7;
                const bf16x8 v8 = *reinterpret_cast<const bf16x8 *>(stg + row * 128 + ((ch ^ (row & 7)) << 4));
                if (i0 + row < Tn) *reinterpret_cast<bf16x8 *>(ctx + ((size_t)b * Tn + i0 + row) * (heads * DHP) + hh * DHP + 8 * ch) = v8;
            }
            return;
        }
    }
    if (dh == DHP) {
        // no padded head dims: straight-line 8-/16-byte stores, rows beyond T go to nowhere by predication only
        if (i0 + il < Tn) {
            T *dst = ctx + ((size_t)b * Tn + i0 + il) * (heads * DHP) + hh * DHP + 4 * g;
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const f32x4 r4 = o[d] * inv;
                if constexpr (sizeof(T) == 2) { bf16x4 w = {(T)r4[0], (T)r4[1], (T)r4[2], (T)r4[3]}; *reinterpret_cast<bf16x4 *>(dst + 16 * d) = w; }
                else { *reinterpret_cast<f32x4 *>(dst + 16 * d) = r4; }
            }
        }
    } else if (i0 + il < Tn) {
        T *dst = ctx + ((size_t)b * Tn + i0 + il) * (heads * dh) + hh * dh;
#pragma unroll
        for (int d = 0; d < DT; ++d) {
            const int dd = 16 * d + 4 * g;
            if (dd + 3 < dh && (dh & 3) == 0) {
                float r4[4] = {o[d][0] * inv, o[d][1] * inv, o[d][2] * inv, o[d][3] * inv};
                if constexpr (sizeof(T) == 2) { bf16x4 w = {(T)r4[0], (T)r4[1], (T)r4[2], (T)r4[3]}; *reinterpret_cast<bf16x4 *>(dst + dd) = w; }
                else { *reinterpret_cast<f32x4 *>(dst + dd) = (f32x4){r4[0], r4[1], r4[2], r4[3]}; }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (dd + r < dh) dst[dd + r] = from_f32<T>(o[d][r] * inv);
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The same computation for lines of at most 320 output frames (the metric's 96x1200 lines: 300), bf16, d_head 64: EVERYTHING A
// (line, head) NEEDS IS RESIDENT IN LDS.  A workgroup (4 waves, one per SIMD, alone on its CU) owns up to 160 queries of one (line, head);
// K, V (all Tk <= 320 keys) and the 16 ntw + Tk rows of P its queries can touch are fetched ONCE by LDS-DMA (143 KB at T = 300), one
// barrier publishes them, and from there on the waves never meet again: no per-key-tile staging round, no barriers in the key loop.
// A wave owns 2 or 3 CONSECUTIVE 16-query tiles and walks the key tiles with all of them at once: the K, V^T and band fragments it reads
// from LDS serve every tile (consecutive query tiles share all but one of their five band tiles per 32 keys), and the tiles' independent
// dependency chains (products -> shift -> maximum -> exponentials -> products) interleave in the one wave a SIMD has.
// The tiled kernel above spends a third of a workgroup's life before its first product and in the staging rounds (cycle stamps: 4.5 k +
// 5 x 0.65 k of 22 k), and each of its five workgroups per (line, head) stages the same K / V rows again.
// Band geometry: local band row lr <-> table row (pos_center - i0 - qmax) + lr, qmax = 16 ntw - 1; score (query i0 + iq, key j) reads
// lr = qmax - iq + j.  For the query tile at local offset iqt and the 32 keys from js: lr = lb + (15 - il + jl), lb = 16 ntw - 16 - iqt + js,
// i.e. the tiled kernel's shift R^T[15 - il + jl][il] on the 48 band rows from lb.
constexpr int AF_QT = 10;                          // query tiles (16 queries) per workgroup at most
constexpr int AF_TK = 320;                         // keys resident at most
constexpr int AF_SK = 20;                          // row stride (floats) of a wave's shift tile
#ifndef COCR_AF_TWOPART
#define COCR_AF_TWOPART 1       // dev: 0 = the first pass staged in one part
#endif
constexpr int AF_WAVES = 8;                        // waves per workgroup: two per SIMD (round 4; the first form had one per SIMD with up to three query tiles)
constexpr int AF_SROWS = 32;                       // rows of a wave's shift tile (the 48 band rows of a step go through it in two overlapping rounds)
constexpr size_t AF_LDS = (size_t)(2 * AF_TK + 16 * AF_QT + AF_TK) * 128 + AF_WAVES * AF_SROWS * AF_SK * sizeof(float);
static_assert(AF_LDS <= 160 * 1024, "LDS of the resident attention kernel");

// A wave's state across the key passes: the query operands of its (at most two) tiles, their output accumulators, denominators and running
// references
struct AttnFullState {
    bf16x8 qu[2][2], qv[2][2];
    f32x4 o[2][4], osum[2];
    float m_run[2];
};

// One key pass: the keys [kg0, kg0 + Tk) of the line are resident in LDS as local rows [0, Tk) (Tk a multiple of 64; kg0 = 0 in the first
// -- for lines of at most 320 frames the only -- pass)
template <int NT>
__device__ __forceinline__ void attn_full_tiles(const unsigned char *ks, const unsigned char *vs, const unsigned char *ps, float *sk,
                                                AttnFullState &st, int lb_first, int Tn, int Tk, int kg0, int lane, bool sync_after_first = false) {
    typedef bf16_t T;
    constexpr int KC = 2, DT = 4, RS = 128;
    const int il = lane & 15, g = lane >> 4;
    const int fo0 = ((g ^ (il & 7)) << 4), fo1 = (((4 + g) ^ (il & 7)) << 4);        // byte offset of this lane's 16 bytes of k-chunk 0 / 1 in a row = il mod 8
    auto &qu = st.qu;
    auto &qv = st.qv;
    auto &o = st.o;
    auto &osum = st.osum;
    auto &m_run = st.m_run;
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = il == 0 ? (T)1.0f : (T)0.0f;

    for (int j0 = 0; j0 < Tk; j0 += 64) {
        const bool first_tile = kg0 + j0 == 0;                                // the line's first key tile sets the running references
        f32x4 sc[NT][4], keep[NT];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int js = j0 + 32 * half;
            // band fragments of the NT + 2 band tiles these query tiles touch for the 32 keys from js: tile u <-> band rows lb_first - 16 (NT - 1) + js + 16 u
            bf16x8 bfr[NT + 2][KC];
            {
                const unsigned char *pb0 = ps + (size_t)(lb_first - 16 * (NT - 1) + js + il) * RS;
#pragma unroll
                for (int u = 0; u < NT + 2; ++u) {
                    if (half == 1 && u == 0) continue;                       // (only query tile NT - 1's kept tile would use it)
                    bfr[u][0] = *reinterpret_cast<const bf16x8 *>(pb0 + u * 16 * RS + fo0);
                    bfr[u][1] = *reinterpret_cast<const bf16x8 *>(pb0 + u * 16 * RS + fo1);
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4 negm = {-m_run[t], -m_run[t], -m_run[t], -m_run[t]};     // positional products start from -reference: see the tiled kernel
#pragma unroll
                for (int ml = 0; ml < 3; ++ml) {
                    f32x4 rr;
                    if (half == 1 && ml == 0) rr = keep[t];                   // band tile 2 of the first half = tile 0 of the second
                    else {
                        rr = negm;
#pragma unroll
                        for (int c = 0; c < KC; ++c) rr = mma16(bfr[ml + (NT - 1 - t)][c], qv[t][c], rr);
                    }
                    if (half == 0 && ml == 2) keep[t] = rr;
                    // The shift tile holds 32 rows: band rows 0..31 are written, the first 16 keys' scores read (rows 15 - il + [0, 16) of them),
                    // then band rows 32..47 overwrite rows 0..15 and the second 16 keys' scores are read (rows 31 - il + [0, 16) mod 32).
                    // No waits around any of it: the tile is private to the wave and a wave's LDS operations execute in order, so every read
                    // sees the writes before it and no write overtakes a read.  The first use of the shifted values, the content products, is
                    // where the wave waits.
#pragma unroll
                    for (int r = 0; r < 4; ++r) sk[((16 * ml + 4 * g + r) & (AF_SROWS - 1)) * AF_SK + il] = rr[r];
                    if (ml >= 1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) sc[t][2 * half + ml - 1][r] = sk[((15 - il + 16 * (ml - 1) + 4 * g + r) & (AF_SROWS - 1)) * AF_SK + il];
                    }
                }
            }
            // content products on top of the shifted positional scores: the K fragments serve every query tile
            bf16x8 kf[2][KC];
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                const unsigned char *kr = ks + (size_t)(js + 16 * tl + il) * RS;
                kf[tl][0] = *reinterpret_cast<const bf16x8 *>(kr + fo0);
                kf[tl][1] = *reinterpret_cast<const bf16x8 *>(kr + fo1);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                    for (int c = 0; c < KC; ++c) sc[t][2 * half + tl] = mma16(kf[tl][c], qu[t][c], sc[t][2 * half + tl]);
        }
        // ---- one softmax step per 64 keys and query tile (sc = score - m_run, log2 units), as in the tiled kernel
        bf16x8 pb[NT][2];
        // The maxima below are inline assembly (max3_f): one fence for all query tiles' scores first
        if constexpr (NT == 1) mfma_results_fence(sc[0]);
        else if constexpr (NT == 2) mfma_results_fence(sc[0], sc[1]);
        else mfma_results_fence(sc[0], sc[1], sc[2]);
        // (the steps below run over all query tiles before the next step starts, and the two uniform branches -- keys beyond the line, reference
        // moved -- are taken once for all tiles: a branch per tile would cut the tiles' chains into basic blocks that cannot interleave)
        float tmax[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float m = max3_f(sc[t][0][0], sc[t][0][1], sc[t][0][2]);
            m = max3_f(m, sc[t][0][3], sc[t][1][0]);
            m = max3_f(m, sc[t][1][1], sc[t][1][2]);
            m = max3_f(m, sc[t][1][3], sc[t][2][0]);
            m = max3_f(m, sc[t][2][1], sc[t][2][2]);
            m = max3_f(m, sc[t][2][3], sc[t][3][0]);
            m = max3_f(m, sc[t][3][1], sc[t][3][2]);
            tmax[t] = max3_f(m, sc[t][3][3], sc[t][3][3]);
        }
        if (kg0 + j0 + 64 > Tn) {                                            // uniform: only the line's last tile has keys beyond T
            asm volatile("; keys beyond the line" ::: "memory");
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float m = -INFINITY;
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (kg0 + j0 + 16 * tt + 4 * g + r >= Tn) sc[t][tt][r] = -INFINITY;
                        m = fmaxf(m, sc[t][tt][r]);
                    }
                tmax[t] = m;
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) { const float x = __shfl_xor(tmax[t], 16, 64); tmax[t] = max3_f(tmax[t], x, x); }
#pragma unroll
        for (int t = 0; t < NT; ++t) { const float x = __shfl_xor(tmax[t], 32, 64); tmax[t] = max3_f(tmax[t], x, x); }
        // Lazy rescaling, per query tile exactly as in the tiled kernel (the reference of a tile moves only when one of ITS queries exceeds it
        // by more than LAZY; the first key tile sets it) -- a tile whose reference stays gets delta = 0, alpha = 2^-0 = 1: the same bits
        constexpr float LAZY = 8.0f;
        bool need[NT], any = first_tile;
#pragma unroll
        for (int t = 0; t < NT; ++t) { need[t] = first_tile || __builtin_amdgcn_ballot_w64(tmax[t] > LAZY) != 0; any = any || need[t]; }
        if (any) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float delta = first_tile ? tmax[t] : (need[t] ? fmaxf(tmax[t], 0.f) : 0.f);
                const float alpha = first_tile ? 1.0f : __builtin_amdgcn_exp2f(-delta);
                m_run[t] = first_tile ? tmax[t] : m_run[t] + delta;
#pragma unroll
                for (int r = 0; r < 4; ++r) osum[t][r] *= alpha;
#pragma unroll
                for (int d = 0; d < DT; ++d)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[t][d][r] *= alpha;
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sc[t][tt][r] -= delta;
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) pb[t][tt >> 1][4 * (tt & 1) + r] = (T)__builtin_amdgcn_exp2f(sc[t][tt][r]);
            osum[t] = mma16(ones, pb[t][0], osum[t]);
            osum[t] = mma16(ones, pb[t][1], osum[t]);
        }
        // ---- O^T += V^T P^T: a V^T fragment serves every query tile
#pragma unroll
        for (int d = 0; d < DT; ++d) {
            const bf16x8 v0 = load_vt_frag<true>(vs, RS, j0, d, il, g, T()), v1 = load_vt_frag<true>(vs, RS, j0 + 32, d, il, g, T());
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                o[t][d] = mma16(v0, pb[t][0], o[t][d]);
                o[t][d] = mma16(v1, pb[t][1], o[t][d]);
            }
        }
        if (j0 == 0 && sync_after_first) {             // (uniform) the rows behind the first key tile were requested while it was computed
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
}

// normalise and store: whole 128-byte rows through the wave's shift tile, as in the tiled kernel
template <int NT>
__device__ __forceinline__ void attn_full_store(float *sk, const AttnFullState &st, bf16_t *ctxh, int ctx_stride, int i_first, int Tn, int lane) {
    typedef bf16_t T;
    constexpr int DT = 4;
    const int il = lane & 15, g = lane >> 4;
    auto &o = st.o;
    auto &osum = st.osum;
    unsigned char *stg = reinterpret_cast<unsigned char *>(sk);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float inv = 1.0f / __shfl(osum[t][0], il, 64);                  // the denominator of query il sits in lane il
#pragma unroll
        for (int d = 0; d < DT; ++d) {
            const f32x4 r4 = o[t][d] * inv;
            const bf16x4 w = {(T)r4[0], (T)r4[1], (T)r4[2], (T)r4[3]};
            *reinterpret_cast<bf16x4 *>(stg + il * 128 + (((2 * d + (g >> 1)) ^ (il & 7)) << 4) + 8 * (g & 1)) = w;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = 8 * j + (lane >> 3), ch = lane & 7;
            const bf16x8 v8 = *reinterpret_cast<const bf16x8 *>(stg + row * 128 + ((ch ^ (row & 7)) << 4));
            const int i = i_first + 16 * t + row;
            if (i < Tn) *reinterpret_cast<bf16x8 *>(ctxh + (size_t)i * ctx_stride + 8 * ch) = v8;
        }
        // (the next tile's staging writes follow these reads in the wave's LDS order)
    }
}

// `kpass`: keys resident per pass (a multiple of 64, at most AF_TK).  Lines of at most kpass frames take one pass (the form described above);
// longer lines (round 4: the wide model's bucketed widths, up to 600 frames and beyond) walk their keys in passes of kpass -- K, V and the
// band of the next pass replace the previous ones between two barriers, the waves' softmax state (AttnFullState) carries over.  Per 320 keys
// and 160 queries a pass stages 143 KB once; the tiled kernel stages 24 KB per (64 keys, 64 queries) = 300 KB for the same pairs, behind two
// barriers per key tile.
template <bool LONG>
__global__ __launch_bounds__(64 * AF_WAVES) void relpos_attention_full_kernel(const bf16_t *__restrict__ q, const bf16_t *__restrict__ k, const bf16_t *__restrict__ v,
                                                                    const bf16_t *__restrict__ ptab, const float *__restrict__ ub,
                                                                    const float *__restrict__ vb, bf16_t *__restrict__ ctx,
                                                                    int Tn, int Tp, int heads, float scale, int pos_center, int ntw, int kpass) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) unsigned char att_smem[];
    const int Tk = (Tn + 63) & ~63;                    // keys walked: whole 64-key tiles; rows beyond T are zero in q / k / v
    const int Kp = min(kpass, Tk);                     // keys of a full pass (rows of K / V staged)
    const int nbr = 16 * ntw + Kp;                     // band rows this workgroup can touch in one pass
    unsigned char *ks = att_smem, *vs = ks + (size_t)Kp * 128, *ps = vs + (size_t)Kp * 128;
    float *skew = reinterpret_cast<float *>(ps + (size_t)nbr * 128);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware order: the workgroups of one (line, head) read the same K / V rows
    const int nqb = gridDim.x, logical = xcd_remap_i(blockIdx.x + nqb * blockIdx.y, nqb * gridDim.y);
    const int qb = logical % nqb, bh = logical / nqb, b = bh / heads, hh = bh - b * heads;
    const int i0 = qb * 16 * ntw;                      // first query of this workgroup
    const int nt_here = min(ntw, (Tn - i0 + 15) >> 4); // its query tiles (the last workgroup of a line may hold fewer)
    // this wave's consecutive query tiles: nt_here (<= 10) tiles over 8 waves, the first (nt_here % 8) waves take one more -- waves w and
    // w + 4 share a SIMD, so at 10 tiles the SIMDs hold 3, 3, 2, 2 tiles in two waves each (one wave's waits are the other's issue slots)
    static_assert(AF_QT <= 2 * AF_WAVES, "at most two query tiles per wave");
    const int base = nt_here / AF_WAVES, extra = nt_here % AF_WAVES;
    const int mine = base + (wave < extra ? 1 : 0), first = wave * base + min(wave, extra);
    const int i_first = i0 + 16 * first;
    // ---- its query rows and the head's two bias vectors: requested ahead of the DMAs below (one wait covers everything)
    bf16x8 qq[2][2];
    f32x4 u4[2][2], v4[2][2];
    {
        const int il = lane & 15, g = lane >> 4;
        const T *qrows = q + (size_t)bh * Tp * 64;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int iq = min(i_first + 16 * min(t, max(mine - 1, 0)) + il, Tn - 1);      // queries beyond T read row T - 1 and store nothing
#pragma unroll
            for (int c = 0; c < 2; ++c) qq[t][c] = load_frag(qrows + (size_t)iq * 64 + c * 32 + 8 * g);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                u4[c][hlf] = *reinterpret_cast<const f32x4 *>(ub + hh * 64 + c * 32 + 8 * g + 4 * hlf);
                v4[c][hlf] = *reinterpret_cast<const f32x4 *>(vb + hh * 64 + c * 32 + 8 * g + 4 * hlf);
            }
    }
    // ---- K, V and the band of the keys [kg0, kg0 + kn) -> LDS by DMA: one wave-instruction = 8 rows of 128 bytes; chunk c of row r lands at
    // chunk c ^ (r & 7).  Band geometry: local band row lr <-> table row (pos_center - i0 - qmax + kg0) + lr, qmax = 16 ntw - 1.
    // `a` .. `b`: the local keys staged by this call (multiples of 64): K / V rows [a, b) and the band rows they add -- [0, 16 ntw + b) for
    // a == 0, [16 ntw + a, 16 ntw + b) behind an earlier call
    auto stage = [&](int kg0, int a, int b2) {
        const int r8 = lane >> 3, cs = lane & 7, gch = (cs ^ r8) * 8;       // this lane's row inside a group of 8 and the GLOBAL chunk it fetches
        const T *kb = k + ((size_t)bh * Tp + kg0) * 64, *vbs = v + ((size_t)bh * Tp + kg0) * 64;
        for (int rg = a / 8 + wave; rg < b2 / 8; rg += AF_WAVES) {
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(kb + (size_t)(rg * 8 + r8) * 64 + gch), (lds_ptr_t)(ks + rg * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(vbs + (size_t)(rg * 8 + r8) * 64 + gch), (lds_ptr_t)(vs + rg * 1024), 16, 0, 0);
        }
        const int prow = heads * 64;
        const T *pb = ptab + (size_t)(pos_center - i0 - (16 * ntw - 1) + kg0) * prow + hh * 64;
        for (int rg = (a ? (16 * ntw + a) / 8 : 0) + wave; rg < (16 * ntw + b2) / 8; rg += AF_WAVES)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(pb + (size_t)(rg * 8 + r8) * prow + gch), (lds_ptr_t)(ps + rg * 1024), 16, 0, 0);
    };
    stage(0, 0, (!LONG && COCR_AF_TWOPART) ? min(64, Kp) : Kp);
    // ---- query operands: (q + u) scale and (q + v) scale of this lane's query in each tile; empty accumulators
    AttnFullState st;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = (float)qq[t][c][j];
                st.qu[t][c][j] = (T)((x + u4[c][j >> 2][j & 3]) * scale);
                st.qv[t][c][j] = (T)((x + v4[c][j >> 2][j & 3]) * scale);
            }
        st.osum[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        st.m_run[t] = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) st.o[t][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int lb_first = 16 * ntw - 16 - 16 * first;   // band row of R^T row 0 for the first tile and the pass's key 0; tile t: 16 t lower; keys from js: + js
    float *sk = skew + wave * AF_SROWS * AF_SK;
    if constexpr (!LONG) {
        // one pass (lines of at most kpass frames; the measured shape).  Staged in two parts: the first key tile (K / V rows 0..63 and the
        // 16 ntw + 64 band rows its scores touch: 44 of 143 KB at the metric's shape) is waited for and published alone, the rest is
        // requested behind that barrier and lands while the tile is computed; a second barrier at the end of the first loop iteration
        // (attn_full_tiles: sync_after_first) publishes it.  Waves without a query tile take part in both barriers and leave.
        const bool two = COCR_AF_TWOPART && Kp > 64;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (two) stage(0, 64, Kp);
        if (mine == 0) {
            if (two) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
            return;
        }
        T *ctxh = ctx + (size_t)b * Tn * (heads * 64) + hh * 64;
        if (mine == 1) { attn_full_tiles<1>(ks, vs, ps, sk, st, lb_first, Tn, Kp, 0, lane, two); attn_full_store<1>(sk, st, ctxh, heads * 64, i_first, Tn, lane); }
        else { attn_full_tiles<2>(ks, vs, ps, sk, st, lb_first, Tn, Kp, 0, lane, two); attn_full_store<2>(sk, st, ctxh, heads * 64, i_first, Tn, lane); }
        return;
    }
    for (int kg0 = 0; kg0 < Tk; kg0 += Kp) {
        const int kn = min(Kp, Tk - kg0);
        if (kg0 > 0) {
            __builtin_amdgcn_s_barrier();              // every wave is done with the previous pass's rows
            stage(kg0, 0, kn);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (mine == 1) attn_full_tiles<1>(ks, vs, ps, sk, st, lb_first, Tn, kn, kg0, lane);
        else if (mine == 2) attn_full_tiles<2>(ks, vs, ps, sk, st, lb_first, Tn, kn, kg0, lane);
    }
    T *ctxh = ctx + (size_t)b * Tn * (heads * 64) + hh * 64;
    if (mine == 1) attn_full_store<1>(sk, st, ctxh, heads * 64, i_first, Tn, lane);
    else if (mine == 2) attn_full_store<2>(sk, st, ctxh, heads * 64, i_first, Tn, lane);
}

template <typename T, int DHP> static inline size_t attention_lds_bytes() {
    const size_t rb = DHP * sizeof(T);
    return (size_t)(64 + 64 + 128) * ((rb == 128 && sizeof(T) == 2) ? rb : rb + 16) + 4 * 48 * 20 * sizeof(float);
}
