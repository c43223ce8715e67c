// Row-local chains of a conformer block in ONE kernel (bf16, encoder_dim 256).
//
// Everything in a block except the attention core (needs all frames of a line) and the depthwise conv (needs +-15 frames)
// maps a row of the (M, D) activation to a row: projections, feed-forward modules, residual adds, LayerNorms.  A workgroup
// that owns 48 rows can therefore run a whole sequence of them back to back with the operand rows staying on chip:
//     chain A (after the attention core):   out-proj + residual + LayerNorm  ->  pointwise conv 1 + GLU
//     chain B (after the depthwise conv):   pointwise conv 2 + residual + LayerNorm  ->  FFN (+ block-final LayerNorm,
//                                            + next block's LayerNorm)  ->  next block's FFN (+ LayerNorm)  ->  its q/k/v projection
// which leaves 4 launches per block (attention core, chain A, depthwise conv, chain B) instead of 8, and removes the ~9 us
// fixed cost (launch, prologue, LayerNorm epilogue) each of the separate kernels paid.  Reference lines: attention.py:70,103
// (out_proj), convolution.py:138-139,143 (pointwise convs, GLU), feed_forward.py:45-52, modules.py:32, encoder.py:62-99.
//
// Machinery (the fused FFN kernel's, generalised): 8 waves side by side; the 48 x 256 operand rows live in registers as MFMA
// fragments; every stage streams its weights in 8 KiB WAVE-PRIVATE slices (16 weight rows x K = 256) through two per-wave
// LDS buffers by LDS-DMA with the wave's own counted vmcnt -- one slice ahead, no workgroup barrier on the weight stream;
// wave w owns 16 of every 128 output columns.  Stage kinds:
//   ROWLN  N = 256:  x <- [x +] alpha (A W^T + b); LayerNorm(s) -> new operand rows (+ x, xn to global when asked)
//   FFN    hidden FF: per 128-wide hidden chunk  H = silu(A W1^T + b1) (LDS, bf16), Y += H W2^T; then as ROWLN
//   GLU    N = 512 packed (value tile, gate tile): out = (a + ba) * sigmoid(g + bg) -> global (M, 256)
//   QKV    N = 768: + bias -> scattered into the attention layouts q, k, v
#pragma once
#include <type_traits>

#include "gemm.hip.h"

enum { ST_ROWLN = 0, ST_FFN = 1, ST_GLU = 2, ST_QKV = 3 };
#define COCR_CHAIN_SMALL_M 4800          // below this many rows the register-streamed chain kernels use 32-row workgroups

struct ChainStage {
    int kind;
    const bf16_t *W;       // ROWLN / GLU / QKV: (N, 256); FFN: W1 (FF, 256)
    const bf16_t *W2;      // FFN: (256, FF)
    const float *bias;     // ROWLN / GLU / QKV: (N); FFN: b1 (FF)
    const float *bias2;    // FFN: b2 (256)
    int N;                 // output columns (256 / 512 / 768) or FF
    float alpha;           // ROWLN / FFN residual factor
    int has_resid;
    const float *g1, *b1, *g2, *b2;    // LayerNorm(s) after the residual add (g2 != null: chained, x <- LN1)
    int store_x, store_xn;             // write the fp32 stream / the normalised operand back to global after this stage
    bf16_t *out;           // GLU: (M, 256)
    bf16_t *q, *k, *v;     // QKV
    float *tap_pre, *tap_post;         // debug taps (TAPS instantiation of chain96_kernel only): (M, 256) fp32 copies of the stream after this
                                       // stage's residual add, and (chained LayerNorms) after the first LayerNorm; null = not wanted
};

struct ChainArgs {
    const bf16_t *A0;      // first operand rows (M, 256)
    float *x;              // fp32 residual stream (M, 256)
    bf16_t *xn;            // normalised operand (M, 256), written when a stage asks for it
    int M, nstages;
    const bf16_t *dw_in;   // 96-row form with the depthwise-conv prologue: GLU output (M, 256); A0 unused
    const float *dw_w, *dw_b;      // BatchNorm-folded depthwise taps [k][256] and bias [256]
    bf16_t *tap_dw;        // debug tap (TAPS instantiation): (M, 256) copy of the depthwise-conv prologue's output, or null
    float *dump;           // >= 16 KiB scratch: branch-free sink for the stores of rows beyond M (16 bytes per thread + slack)
    unsigned long long *stamps;   // dev: cycle stamps of workgroup 0 / wave 0 (COCR_CHAIN_STAMPS), else null
    int dh, dhp, heads, T_, Tp;       // attention layout of the QKV stage
    ChainStage st[4];
};

template <int K0, int K1, int K2, int K3>     // stage kinds, -1 = none: one specialised kernel per chain shape
__global__ __launch_bounds__(512) void chain_kernel(ChainArgs p) {
    typedef bf16_t T;
    constexpr int D = 256, BMC = 48, KC1 = D / 32;
    constexpr int HPANEL = BMC * 128;           // one [48 rows][128 B] panel
    constexpr int WBUF = 8192;                  // one weight slice: [4 panels][16 rows][128 B] (or [2 panels][32 rows][128 B] for FFN W2)
    constexpr int RS = D * 4 + 16;              // fp32 staged row
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *hs = smem;                               // 4 panels: operand tile / 2 hidden-chunk buffers / bf16 output staging
    unsigned char *wreg = hs + 4 * HPANEL;                  // 8 waves x 2 slices; the fp32 LayerNorm staging reuses it between stages
    float *b1s = reinterpret_cast<float *>(wreg + 8 * 2 * WBUF);   // FFN b1 (FF floats, up to 4 KiB... sized by the launcher)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4, swz = r16 & 7;
    const int M = p.M, m0 = blockIdx.x * BMC, mend = min(M, m0 + BMC);
    const int lrow = lane >> 3, cpos = lane & 7;
    unsigned char *wbuf[2] = {wreg + wave * 2 * WBUF, wreg + wave * 2 * WBUF + WBUF};

    // ---- residual rows of this wave (row groups `wave` and `wave + 8` of 12) and the first operand tile
    const int rg0 = 4 * wave, rg1 = 4 * min(wave + 8, 11);
    EpiResidualLN<T, 1> ld{p.x, D, nullptr, 1.f, D, 1, nullptr, nullptr, nullptr, nullptr, nullptr};
    EpiResidualLN<T, 1>::Rows4 xr[2];
    xr[0] = ld.rows4_load(m0 + rg0, mend, lane);
    xr[1] = ld.rows4_load(m0 + rg1, mend, lane);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int id = wave + 8 * i, pnl = id / 6, rg = id - pnl * 6, row = rg * 8 + lrow;
        const T *src = p.A0 + (size_t)min(m0 + row, M - 1) * D + pnl * 64 + ((cpos ^ (row & 7)) * 8);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(hs + pnl * HPANEL + rg * 1024), 16, 0, 0);
    }
    bf16x8 xa[3][KC1];
    auto load_operand = [&]() {          // hs (4 panels, swizzled A image) -> fragments; callers fence before and after
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int kk = 0; kk < KC1; ++kk) xa[i][kk] = lds_frag_swz(hs + (kk >> 1) * HPANEL + (16 * i + r16) * 128, kk & 1, g, swz, T());
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    load_operand();
    __builtin_amdgcn_s_barrier();

    // ---- weight slices: 8 DMA wave-instructions each
    auto issue16 = [&](const T *Wbase, int ld, int row0, unsigned char *buf) {      // 16 rows x 256 k  -> [4 panels][16 rows]
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pnl = i >> 1, row = (i & 1) * 8 + lrow;
            const T *src = Wbase + (size_t)(row0 + row) * ld + pnl * 64 + ((cpos ^ (row & 7)) * 8);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(buf + pnl * 2048 + (i & 1) * 1024), 16, 0, 0);
        }
    };
    auto issue32 = [&](const T *Wbase, int ld, int row0, int k0, unsigned char *buf) {   // 32 rows x 128 k -> [2 panels][32 rows]
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pnl = i >> 2, row = (i & 3) * 8 + lrow;
            const T *src = Wbase + (size_t)(row0 + row) * ld + k0 + pnl * 64 + ((cpos ^ (row & 7)) * 8);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(buf + pnl * 4096 + (i & 3) * 1024), 16, 0, 0);
        }
    };
    auto mma_rows16 = [&](const unsigned char *buf, int row_off, f32x4 (&acc)[3]) {  // acc[48 x 16] += operand . slice rows [row_off, +16)
#pragma unroll
        for (int kk = 0; kk < KC1; ++kk) {
            const bf16x8 b = lds_frag_swz(buf + (kk >> 1) * 2048 + (row_off + r16) * 128, kk & 1, g, swz, T());
#pragma unroll
            for (int i = 0; i < 3; ++i) acc[i] = mma16(b, xa[i][kk], acc[i]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads of the slice retired before it is refilled
    };
    // cooperative, coalesced copy of a bf16 [48][128] tile in hs (row stride 256 B at `tile`) to global through `store`
    auto flush_tile = [&](const unsigned char *tile, auto &&store) {
        for (int id = tid; id < BMC * 16; id += 512) {
            const int row = id >> 4, ch = id & 15;
            if (m0 + row < mend) store(m0 + row, ch * 8, reinterpret_cast<const T *>(tile + row * 256 + ch * 16));
        }
    };

    // ---- residual + LayerNorm epilogue on a staged fp32 [48][256] tile at wreg; leaves the new operand tile in hs and in xa
    auto rowln_epilogue = [&](const ChainStage &st) {
        EpiResidualLN<T, 1> e{p.x, D, nullptr, 1.f, D, st.has_resid, st.g1, st.b1, st.g2, st.b2, p.xn};
        __syncthreads();                                     // staged tile complete
        e.rows4_chain(m0 + rg0, mend, reinterpret_cast<const float *>(wreg + rg0 * RS), RS / 4, lane, xr[0], st.store_x, st.store_xn, hs, rg0, HPANEL);
        if (wave < 4)
            e.rows4_chain(m0 + rg1, mend, reinterpret_cast<const float *>(wreg + rg1 * RS), RS / 4, lane, xr[1], st.store_x, st.store_xn, hs, rg1, HPANEL);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                     // new operand tile complete, staging consumed
        load_operand();
        __syncthreads();                                     // hs free again (hidden chunks / output staging)
    };

    auto run_stage = [&](auto KIND, const ChainStage &st) {
        constexpr int kind = decltype(KIND)::value;
        if constexpr (kind == ST_ROWLN) {
            // two 128-column chunks; wave w owns columns c*128 + 16w .. +15
            issue16(st.W, D, 16 * wave, wbuf[0]);
            issue16(st.W, D, 128 + 16 * wave, wbuf[1]);
            f32x4 acc[2][3];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#pragma unroll
                for (int i = 0; i < 3; ++i) acc[c][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (c == 0) wait_vmcnt<8>(); else wait_vmcnt<0>();
                mma_rows16(wbuf[c], 0, acc[c]);
            }
            __syncthreads();                                 // every wave is done with its slices: the area becomes the fp32 staging
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int row = 16 * i + r16, n = c * 128 + 16 * wave + 4 * g;
                    const f32x4 bb = *reinterpret_cast<const f32x4 *>(st.bias + n);
                    f32x4 r;
#pragma unroll
                    for (int q = 0; q < 4; ++q) r[q] = st.alpha * (acc[c][i][q] + bb[q]);
                    *reinterpret_cast<f32x4 *>(wreg + row * RS + n * 4) = r;
                }
            rowln_epilogue(st);
        } else if constexpr (kind == ST_FFN) {
            const int FF = st.N, nchunks = FF / 128;
            for (int i = wave; i < FF / 256; i += 8)
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(st.bias + i * 256 + lane * 4), (lds_ptr_t)(reinterpret_cast<unsigned char *>(b1s) + i * 1024), 16, 0, 0);
            issue16(st.W, D, 16 * wave, wbuf[0]);                         // W1(0)
            issue32(st.W2, FF, 32 * wave, 0, wbuf[1]);                    // W2(0)
            wait_vmcnt<16>();
            __syncthreads();                                              // b1 in LDS for everyone
            f32x4 acc2[3][2];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int c = 0; c < nchunks; ++c) {
                wait_vmcnt<8>();                                          // W1(c) landed, W2(c) may be in flight
                f32x4 acc1[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) acc1[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                mma_rows16(wbuf[0], 0, acc1);
                if (c + 1 < nchunks) issue16(st.W, D, (c + 1) * 128 + 16 * wave, wbuf[0]);
                {
                    unsigned char *hb = hs + (c & 1) * 2 * HPANEL;
                    const int jj = 16 * wave + 4 * g;
                    const f32x4 bb = *reinterpret_cast<const f32x4 *>(b1s + c * 128 + jj);
                    const int ch16 = (jj & 63) >> 3;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const int row = 16 * i + r16;
                        bf16x4 hv;
#pragma unroll
                        for (int q = 0; q < 4; ++q) hv[q] = (T)silu_f(acc1[i][q] + bb[q]);
                        *reinterpret_cast<bf16x4 *>(hb + (jj >> 6) * HPANEL + row * 128 + ((ch16 ^ (row & 7)) << 4) + ((jj & 7) >> 2) * 8) = hv;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                             // hidden chunk complete
                if (c + 1 < nchunks) wait_vmcnt<8>(); else wait_vmcnt<0>();   // W2(c) landed
                const unsigned char *hb = hs + (c & 1) * 2 * HPANEL;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    bf16x8 a[3], b[2];
#pragma unroll
                    for (int i = 0; i < 3; ++i) a[i] = lds_frag_swz(hb + (kk >> 1) * HPANEL + (16 * i + r16) * 128, kk & 1, g, swz, T());
#pragma unroll
                    for (int j = 0; j < 2; ++j) b[j] = lds_frag_swz(wbuf[1] + (kk >> 1) * 4096 + (16 * j + r16) * 128, kk & 1, g, swz, T());
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc2[i][j] = mma16(b[j], a[i], acc2[i][j]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (c + 1 < nchunks) issue32(st.W2, FF, 32 * wave, (c + 1) * 128, wbuf[1]);
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int row = 16 * i + r16, n = 32 * wave + 16 * j + 4 * g;
                    const f32x4 bb = *reinterpret_cast<const f32x4 *>(st.bias2 + n);
                    f32x4 r;
#pragma unroll
                    for (int q = 0; q < 4; ++q) r[q] = st.alpha * (acc2[i][j][q] + bb[q]);
                    *reinterpret_cast<f32x4 *>(wreg + row * RS + n * 4) = r;
                }
            rowln_epilogue(st);
        } else if constexpr (kind == ST_GLU) {
            // two steps of 256 packed columns; wave w owns packed rows step*256 + 32w .. +31 = (value tile, gate tile) of
            // channels step*128 + 16w .. +15
            EpiGLU<T> e{st.out, D, st.bias, 2 * D};
            issue16(st.W, D, 32 * wave, wbuf[0]);
            issue16(st.W, D, 32 * wave + 16, wbuf[1]);
            for (int step = 0; step < 2; ++step) {
                f32x4 av[3], ag[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) { av[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; ag[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
                wait_vmcnt<0>();
                mma_rows16(wbuf[0], 0, av);
                mma_rows16(wbuf[1], 0, ag);
                if (step == 0) { issue16(st.W, D, 256 + 32 * wave, wbuf[0]); issue16(st.W, D, 256 + 32 * wave + 16, wbuf[1]); }
                unsigned char *tile = hs + step * 2 * HPANEL;            // bf16 [48][128], row stride 256 B
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int row = 16 * i + r16, n = step * 256 + 32 * wave + 4 * g;      // packed column of the value tile
                    const float v[4] = {av[i][0], av[i][1], av[i][2], av[i][3]}, w[4] = {ag[i][0], ag[i][1], ag[i][2], ag[i][3]};
                    float r[4];
                    e.transform(n, v, w, r);
                    bf16x4 o = {(T)r[0], (T)r[1], (T)r[2], (T)r[3]};
                    *reinterpret_cast<bf16x4 *>(tile + row * 256 + (16 * wave + 4 * g) * 2) = o;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                flush_tile(tile, [&](int m, int c, const T *src) { e.store(m, step * 128 + c, src, 8); });
            }
        } else if constexpr (kind == ST_QKV) {   // six 128-column chunks
            EpiQKV<T> e{st.q, st.k, st.v, st.bias, D, p.dh, p.dhp, p.heads, p.T_, p.Tp, 3 * D};
            issue16(st.W, D, 16 * wave, wbuf[0]);
            for (int c = 0; c < 6; ++c) {
                if (c + 1 < 6) issue16(st.W, D, (c + 1) * 128 + 16 * wave, wbuf[(c + 1) & 1]);
                if (c + 1 < 6) wait_vmcnt<8>(); else wait_vmcnt<0>();
                f32x4 acc[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                mma_rows16(wbuf[c & 1], 0, acc);
                unsigned char *tile = hs + (c & 1) * 2 * HPANEL;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int row = 16 * i + r16, n = c * 128 + 16 * wave + 4 * g;
                    const float v[4] = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
                    float r[4];
                    e.transform(n, v, r);
                    bf16x4 o = {(T)r[0], (T)r[1], (T)r[2], (T)r[3]};
                    *reinterpret_cast<bf16x4 *>(tile + row * 256 + (16 * wave + 4 * g) * 2) = o;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                             // tile c complete (tile c-1 was flushed before this barrier)
                flush_tile(tile, [&](int m, int cc, const T *src) { e.store(m, c * 128 + cc, src, 8); });
            }
        }
    };
    run_stage(std::integral_constant<int, K0>{}, p.st[0]);
    run_stage(std::integral_constant<int, K1>{}, p.st[1]);
    run_stage(std::integral_constant<int, K2>{}, p.st[2]);
    run_stage(std::integral_constant<int, K3>{}, p.st[3]);
}

static inline bool chain_supported(int D, int FF, int dh) { return D == 256 && FF % 256 == 0 && FF >= 256 && FF <= 1024 && dh % 8 == 0; }

template <int K0, int K1, int K2, int K3>
static inline hipError_t launch_chain_cfg(hipStream_t s, const ChainArgs &a, int max_ff) {
    const size_t lds = (size_t)4 * 48 * 128 + 8 * 2 * 8192 + (size_t)max_ff * 4 + 64;
    auto kern = chain_kernel<K0, K1, K2, K3>;
    hipError_t e = raise_lds_limit((const void *)kern, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(ceil_div(a.M, 48)), dim3(512), lds, s, a);
    return hipGetLastError();
}

// the chain shapes the forward uses
static inline hipError_t launch_chain(hipStream_t s, const ChainArgs &a, int max_ff) {
    const int k0 = a.st[0].kind, k1 = a.nstages > 1 ? a.st[1].kind : -1, k2 = a.nstages > 2 ? a.st[2].kind : -1, k3 = a.nstages > 3 ? a.st[3].kind : -1;
    if (k0 == ST_FFN && k1 == ST_QKV && k2 == -1) return launch_chain_cfg<ST_FFN, ST_QKV, -1, -1>(s, a, max_ff);
    if (k0 == ST_ROWLN && k1 == ST_GLU && k2 == -1) return launch_chain_cfg<ST_ROWLN, ST_GLU, -1, -1>(s, a, max_ff);
    if (k0 == ST_ROWLN && k1 == ST_FFN && k2 == ST_FFN && k3 == ST_QKV) return launch_chain_cfg<ST_ROWLN, ST_FFN, ST_FFN, ST_QKV>(s, a, max_ff);
    if (k0 == ST_ROWLN && k1 == ST_FFN && k2 == -1) return launch_chain_cfg<ST_ROWLN, ST_FFN, -1, -1>(s, a, max_ff);
    return hipErrorInvalidValue;
}


// =====================================================================================================================
// 96-row form: weights stream L2 -> REGISTERS, activations stay in LDS.
//
// A CU takes in ~70 GB/s from L2 whatever the instruction (MI355X_MICROARCH.md, "Indexed rows"); a 48-row workgroup streams
// every weight byte of the chain for 48 rows = 48 flop per ingested byte, a third of what the CU's MFMA rate needs, and the
// 48-row kernel above is bound by exactly that.  This form doubles the rows per workgroup (the LDS that held the weight
// slices now holds a 96-row operand image), so the same weight stream feeds twice the arithmetic:
//   * weights are read from a FRAGMENT-MAJOR copy (pack_frag_kernel: [n/32][k/32][2][64 lanes][8]) so that the 16 MFMA
//     B-fragments a wave consumes in one step (32 output columns x K = 256) are one contiguous 16 KiB run, each wave
//     instruction a fully coalesced 1 KiB read straight into the registers the MFMA reads;
//   * a 16-fragment register ring per wave: fragment (kk, j) of the NEXT step is requested right after the MFMAs that consumed
//     fragment (kk, j) of this step -- 16 KiB per wave (128 KiB per CU) in flight at all times, across step, stage and
//     epilogue boundaries; the compiler's vmcnt bookkeeping orders it (no LDS-DMA, no hand-counted waits);
//   * the operand rows (96 x 256 bf16, 48 KiB swizzled image) and the FFN hidden chunk (96 x 256, double buffered) are MFMA
//     A-operands read from LDS, 6 row tiles against 2 column tiles per k-step (12 MFMAs per 6 ds_read_b128);
//   * the fp32 residual rows are re-read from global by the thread that wrote them (L2 hits; same-thread program order),
//     instead of living in 48 VGPRs.
// 9600 rows = 100 workgroups: one launch fills 100 CUs, a second stream's launch runs beside it.
// =====================================================================================================================
__global__ __launch_bounds__(256) void pack_frag_kernel(const bf16_t *__restrict__ src, bf16_t *__restrict__ dst, int N, int K) {
    const size_t units = (size_t)N * K / 8;
    for (size_t u = (size_t)blockIdx.x * 256 + threadIdx.x; u < units; u += (size_t)gridDim.x * 256) {
        const int lane = (int)(u & 63), j = (int)((u >> 6) & 1);
        const size_t blk = u >> 7;                        // (pair, kt)
        const int kt = (int)(blk % (K / 32)), pair = (int)(blk / (K / 32));
        const int n = pair * 32 + j * 16 + (lane & 15), k = kt * 32 + 8 * (lane >> 4);
        *reinterpret_cast<bf16x8 *>(dst + u * 8) = *reinterpret_cast<const bf16x8 *>(src + (size_t)n * K + k);
    }
}

// sum over each 16-lane row of the wave (4 DPP steps), result in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {      // (every lane is written by these patterns: `old` = v spares a v_mov per step)
    v += dpp_f32<0xB1, 0xF>(v, v);        // quad_perm(1,0,3,2)
    v += dpp_f32<0x4E, 0xF>(v, v);        // quad_perm(2,3,0,1)
    v += dpp_f32<0x141, 0xF>(v, v);       // row_half_mirror
    v += dpp_f32<0x140, 0xF>(v, v);       // row_mirror
    return v;
}

// DWK != 0: the chain starts with the conv module's depthwise conv (kernel DWK, zero padding, BatchNorm folded) + SiLU on the
// GLU output (convolution.py:140-142), computed for the workgroup's 96 rows from a (96 + DWK - 1)-row window in LDS, result
// written straight into the operand image -- one launch and one (M, 256) round trip less per block.
// MT = 16-row tiles per workgroup: 6 (96 rows: the throughput form) or 2 (32 rows: three times the workgroups for small
// batches, where the latency of one workgroup's serial chain is the forward's latency).
// TAPS: the debug instantiation (cocr_set_debug): the SAME code plus copies of the values that otherwise never leave the chip or are
// overwritten inside the launch (stream after each residual add / first LayerNorm, depthwise output) into the stages' tap buffers.
template <int MT, int DWK, int K0, int K1, int K2, int K3, bool TAPS = false>
__global__ __launch_bounds__(512) void chain96_kernel(ChainArgs p) {
    typedef bf16_t T;
    static_assert(MT == 6 || MT == 2, "rows per wave in the LayerNorm epilogue must be a multiple of 4");
    constexpr int D = 256, BMC = 16 * MT, KC1 = D / 32;
    constexpr int RGH = BMC / 16;              // operand-tile DMA: (4 panels x BMC/8 row groups) / 8 waves wave-instructions per wave
    constexpr int LNP = BMC / 32;              // LayerNorm epilogue: passes of 4 rows per wave (BMC / 8 rows per wave)
    constexpr int PANEL = BMC * 128;            // one [96 rows][128 B] panel of an operand image (64 bf16 of k per row)
    constexpr int IMG = 4 * PANEL;              // 96 x 256 bf16
    constexpr int RS = D * 4 + 16;              // fp32 staged row (LayerNorm epilogue)
    constexpr int OS = D * 2 + 16;              // bf16 staged row (GLU / QKV output tiles)
    constexpr int SLICE = 16 * 512;             // elements in one step's weight run (16 fragments)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *xa = smem;                   // operand image
    unsigned char *hs = smem + IMG;             // 2 hidden-chunk images; fp32 staging (96 * RS) / 2 output tiles (2 * 96 * OS) alias it (+ 4 KiB slack)
    const float *lnp = reinterpret_cast<const float *>(smem + 3 * IMG + 4096);   // LayerNorm parameters of the running stage: g1, b1, g2, b2 (4 x 256 floats)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4, swz = r16 & 7;
    const int M = p.M, m0 = blockIdx.x * BMC, mend = min(M, m0 + BMC);
    const int lrow = lane >> 3, cpos = lane & 7;
    auto lds_fence_barrier0 = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    constexpr int DWPAD = DWK ? (DWK - 1) / 2 : 0, DWROWS = BMC + 2 * DWPAD;
    if constexpr (DWK == 0) {
        // ---- first operand tile -> LDS image: 4 panels x 12 row groups of 8 rows = 48 wave-instructions (6 per wave)
#pragma unroll
        for (int i = 0; i < RGH; ++i) {
            const int id = wave + 8 * i, pnl = id / (BMC / 8), rg = id - pnl * (BMC / 8), row = rg * 8 + lrow;
            const T *src = p.A0 + (size_t)min(m0 + row, M - 1) * D + pnl * 64 + ((cpos ^ (row & 7)) * 8);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(xa + pnl * PANEL + rg * 1024), 16, 0, 0);
        }
    } else {
        // ---- depthwise window: rows m0 - PAD .. m0 + 96 + PAD - 1 of the GLU output (addresses clamped; frames outside the
        // row's own line are excluded by the tap range below), [row][512 B] at hs: one wave-instruction = 2 rows
        static_assert(DWROWS % 2 == 0, "window rows are loaded in pairs");
        for (int q2 = wave; q2 < DWROWS / 2; q2 += 8) {
            const int j = 2 * q2 + (lane >> 5), mrow = min(max(m0 - DWPAD + j, 0), M - 1);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(p.dw_in + (size_t)mrow * D + (lane & 31) * 8), (lds_ptr_t)(hs + q2 * 1024), 16, 0, 0);
        }
    }
    // depthwise taps and bias of this thread's channel pair: requested with the window, ahead of the weight ring (loads return in
    // order: behind the ring they would wait for all of it, and a second time for their own round trip after the barrier)
    typedef float dw_f32x2 __attribute__((ext_vector_type(2)));
    dw_f32x2 dw_wt[DWK ? DWK : 1], dw_bias = {0.f, 0.f};
    if constexpr (DWK != 0) {
        const int c = 2 * (tid & 127);
#pragma unroll
        for (int tau = 0; tau < DWK; ++tau) dw_wt[tau] = *reinterpret_cast<const dw_f32x2 *>(p.dw_w + (size_t)tau * D + c);
        dw_bias = *reinterpret_cast<const dw_f32x2 *>(p.dw_b + c);
    }
    // ---- weight ring: the first step's 16 fragments
    bf16x8 ring[16];
    auto fill = [&](const T *slice, int f) { ring[f] = *reinterpret_cast<const bf16x8 *>(slice + f * 512 + lane * 8); };
    {
        const T *first = p.st[0].W + (size_t)wave * SLICE;
#pragma unroll
        for (int f = 0; f < 16; ++f) fill(first, f);
    }
    // row -> offset of its (batch, frame) in the q / k / v layouts [B][h][Tp][dhp] (the q/k/v stage's scatter would otherwise
    // divide per 16-byte store)
    long long *rowoff = reinterpret_cast<long long *>(smem + 3 * IMG + 8192);
    int *tpos = reinterpret_cast<int *>(smem + 3 * IMG + 8192 + 768);        // frame index of each row inside its line
    if (tid < BMC) {
        const int m = min(m0 + tid, M - 1), b = m / p.T_, t = m - b * p.T_;
        rowoff[tid] = ((long long)b * p.heads * p.Tp + t) * p.dhp;
        tpos[tid] = t;
    }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");        // the operand DMAs (older than the 16 ring loads) have landed
    lds_fence_barrier0();

#ifdef COCR_CHAIN_STAMPS_BUILD                   // dev: cycle stamps of workgroup 0 / thread 0 at the phase boundaries
    int nstamp = 0;
    auto stamp = [&]() {
        if (p.stamps && blockIdx.x == 0 && tid == 0) p.stamps[nstamp] = __builtin_readcyclecounter();
        ++nstamp;
    };
#else
    auto stamp = [&]() {};
#endif
    stamp();
    if constexpr (DWK != 0) {
        // thread = one channel pair x 24 rows (3 groups of 8); a wave's lanes share their rows, so the boundary test is uniform.
        // Accumulation order per output: bias, then taps ascending (as the stand-alone kernel).
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
        const int cp = tid & 127, c = 2 * cp, rq = __builtin_amdgcn_readfirstlane(tid >> 7);
        const dw_f32x2 (&wt)[DWK ? DWK : 1] = dw_wt;
        const f32x2 bias = dw_bias;
        const int T_ = p.T_;
#pragma unroll 1
        for (int grp = 0; grp < BMC / 32; ++grp) {
            const int r0 = (BMC / 4) * rq + 8 * grp;
            const int t0 = __builtin_amdgcn_readfirstlane(tpos[r0]);
            f32x2 acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = bias;
            const unsigned char *wbase = hs + (size_t)r0 * 512 + c * 2;
            if (t0 >= DWPAD && t0 + 7 + DWPAD < T_) {              // all 8 rows inside one line, full tap range
#pragma unroll
                for (int rin = 0; rin < 8 + DWK - 1; ++rin) {
                    const bf16x2 xv = *reinterpret_cast<const bf16x2 *>(wbase + rin * 512);
                    const f32x2 xf = {(float)xv[0], (float)xv[1]};
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int tau = rin - i;
                        if (tau >= 0 && tau < DWK) acc[i] = __builtin_elementwise_fma(wt[tau], xf, acc[i]);
                    }
                }
            } else {                                               // near a line end: tap tau of row i is in range iff 0 <= t_i + tau - PAD < T
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int ti = __builtin_amdgcn_readfirstlane(tpos[r0 + i]);
#pragma unroll
                    for (int tau = 0; tau < DWK; ++tau) {
                        const bf16x2 xv = *reinterpret_cast<const bf16x2 *>(wbase + (i + tau) * 512);
                        const bool ok = (unsigned)(ti + tau - DWPAD) < (unsigned)T_;
                        const f32x2 xf = {ok ? (float)xv[0] : 0.f, ok ? (float)xv[1] : 0.f};
                        acc[i] = __builtin_elementwise_fma(wt[tau], xf, acc[i]);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = r0 + i;
                const bf16x2 o = {(T)silu_f(acc[i][0]), (T)silu_f(acc[i][1])};
                *reinterpret_cast<bf16x2 *>(xa + (c >> 6) * PANEL + row * 128 + ((((c & 63) >> 3) ^ (row & 7)) << 4) + (c & 7) * 2) = o;
                if constexpr (TAPS) {
                    if (p.tap_dw && m0 + row < mend) *reinterpret_cast<bf16x2 *>(p.tap_dw + (size_t)(m0 + row) * D + c) = o;
                }
            }
        }
        stamp();
        lds_fence_barrier0();                                // operand image complete; the window (hs) is free
        stamp();
    }
    // One step: acc[96 rows][32 columns of this wave] += image . ring ; ring <- the 16 fragments at `nxt`.  `side(kk)` is
    // independent VALU work folded into the k-step (the MFMA pipe runs beside it).
    // `fresh`: the accumulators start from zero -- the first k-step's MFMAs take the constant 0 as their C operand instead of 48
    // registers zeroed by 48 v_mov (the VALU is the scarce unit of this kernel).
    auto step = [&](const unsigned char *img, f32x4 (&acc)[MT][2], const T *nxt, auto &&side, auto FRESH) {
        constexpr bool fresh = decltype(FRESH)::value;
#pragma unroll
        for (int kk = 0; kk < KC1; ++kk) {
            // row tiles in two halves: 12 operand registers live instead of 24 (the 96-row FFN stage is at the 256-register limit)
            constexpr int HT = MT > 3 ? MT / 2 : MT;
#pragma unroll
            for (int h0 = 0; h0 < MT; h0 += HT) {
                bf16x8 a[HT];
#pragma unroll
                for (int i = 0; i < HT; ++i) a[i] = lds_frag_swz(img + (kk >> 1) * PANEL + (16 * (h0 + i) + r16) * 128, kk & 1, g, swz, T());
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < HT; ++i)
                        acc[h0 + i][j] = mma16(ring[2 * kk + j], a[i], (fresh && kk == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[h0 + i][j]);
            }
            fill(nxt, 2 * kk);
            fill(nxt, 2 * kk + 1);
            side(std::integral_constant<int, 0>{}, kk);
            __builtin_amdgcn_sched_barrier(0);               // keep the refill (and the side work) here: the scheduler otherwise sinks all of it to the step's end
        }
    };
    auto no_side = [](auto, int) {};
    auto zero = [&](f32x4 (&acc)[MT][2]) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    auto lds_fence_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    // LayerNorm parameters of a stage -> LDS, requested at the stage's start by waves 0..3 (1 KiB each); the epilogue's
    // vmcnt(16) + barrier publishes them (at least one step = 16 younger ring loads lies between)
    auto request_ln_params = [&](const ChainStage &st) {
        if (wave < 4) {
            const float *src = wave == 0 ? st.g1 : wave == 1 ? st.b1 : wave == 2 ? (st.g2 ? st.g2 : st.g1) : (st.b2 ? st.b2 : st.b1);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + lane * 4), (lds_ptr_t)(smem + 3 * IMG + 4096 + wave * 1024), 16, 0, 0);
        }
    };
    // this wave's 2 x 4 bias values of a 256-column product (requested BEFORE the step: younger loads than the ring's would
    // make their consumer wait for the whole ring)
    struct Bias2 { f32x4 v[2]; };
    auto load_bias = [&](const float *bias) {
        Bias2 b;
        b.v[0] = *reinterpret_cast<const f32x4 *>(bias + 32 * wave + 4 * g);
        b.v[1] = *reinterpret_cast<const f32x4 *>(bias + 32 * wave + 16 + 4 * g);
        return b;
    };
    // alpha (acc + bias) of a 256-column product -> fp32 staging rows
    auto stage_rows = [&](const f32x4 (&acc)[MT][2], const Bias2 &bb, float alpha) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = 32 * wave + 16 * j + 4 * g;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                f32x4 r;
#pragma unroll
                for (int q = 0; q < 4; ++q) r[q] = alpha * (acc[i][j][q] + bb.v[j][q]);
                *reinterpret_cast<f32x4 *>(hs + (16 * i + r16) * RS + n * 4) = r;
            }
        }
    };
    // Residual + LayerNorm(s) on the staged tile.  Wave w owns rows 12w .. 12w+11, four at a time: a 16-lane row of the wave
    // holds one activation row (lane c of it: the float4 chunks c, c+16, c+32, c+48), so the four rows reduce together in
    // four row-local DPP steps.  x goes back to global (it is this chain's residual storage, re-read by the same thread in
    // the next epilogue); the normalised rows become the new operand image.
    //   single:  x <- x + staged ;       xn <- LN1(x)
    //   chained: x <- LN1(x + staged) ;  xn <- LN2(x)          (block-final LayerNorm + the next block's first)
    // The 12 residual rows of this wave's share of the LayerNorm epilogue.  Requested BEFORE the stage's last product step (its
    // 16 ring refills are then younger: the epilogue waits for these loads only, and their ~2 us of latency pass under the step).
    // The last pass's rows are requested at the START of the epilogue instead (LATE = 1 at 96 rows): 16 registers less across the
    // product step (the 96-row FFN form sits at the 256-register limit; together with the halved operand fragments in `step` the
    // dominant kernel's scratch went 64 -> 36 bytes per lane; time unchanged within the +-2 us run-to-run noise).
    constexpr int LATE = LNP >= 3 ? 1 : 0, EARLY = LNP - LATE;
    struct Resid { f32x4 v[EARLY][4]; };
    auto resid_row = [&](const ChainStage &st, int pass, f32x4 (&out)[4]) {
        const int rl = lane >> 4, cl = lane & 15;
        const int m = min(m0 + (BMC / 8) * wave + 4 * pass + rl, mend - 1);
#pragma unroll
        for (int v = 0; v < 4; ++v)
            out[v] = *reinterpret_cast<const f32x4 *>(p.x + (size_t)m * D + 4 * (cl + 16 * v));      // (every stage of these chains has a residual: no branch in front of the product step)
    };
    auto load_resid = [&](const ChainStage &st) {
        Resid r;
#pragma unroll
        for (int pass = 0; pass < EARLY; ++pass) resid_row(st, pass, r.v[pass]);
        return r;
    };
    auto rowln_epilogue = [&](const ChainStage &st, const Resid &res) {
        const int rl = lane >> 4, cl = lane & 15;
        const bool chained = st.g2 != nullptr;
        if (wave < 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // this stage's LayerNorm parameters have landed
        lds_fence_barrier();                                 // staged tile complete; every wave is done reading the old image
        f32x4 xr[LNP][4];
#pragma unroll
        for (int pass = 0; pass < EARLY; ++pass)
#pragma unroll
            for (int v = 0; v < 4; ++v) xr[pass][v] = res.v[pass][v];
#pragma unroll
        for (int pass = EARLY; pass < LNP; ++pass) resid_row(st, pass, xr[pass]);
        constexpr float inv_d = 1.0f / (float)D;
        auto normalise = [&](f32x4 (&t)[4], int which) {     // which: 0 = (g1, b1), 1 = (g2, b2)
            const float *ga = lnp + which * 512, *be = ga + 256;
            // packed arithmetic throughout (the epilogue is VALU-issue bound): 4-wide partial sums, one horizontal add each
            f32x4 s4 = (t[0] + t[1]) + (t[2] + t[3]);
            const float mean = row16_sum((s4[0] + s4[1]) + (s4[2] + s4[3])) * inv_d;
            const f32x4 m4 = {mean, mean, mean, mean};
            f32x4 q4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int v = 0; v < 4; ++v) { t[v] -= m4; q4 += t[v] * t[v]; }
            const float rstd = __builtin_amdgcn_rsqf(row16_sum((q4[0] + q4[1]) + (q4[2] + q4[3])) * inv_d + 1e-5f);
            const f32x4 r4 = {rstd, rstd, rstd, rstd};
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const f32x4 gv = *reinterpret_cast<const f32x4 *>(ga + 4 * (cl + 16 * v)), bv = *reinterpret_cast<const f32x4 *>(be + 4 * (cl + 16 * v));
                t[v] = t[v] * (r4 * gv) + bv;
            }
        };
        // One basic block for all passes (no branch inside: rows beyond M and the optional xn copy store through selected addresses
        // into a dump area), so that the scheduler interleaves the passes' dependent chains (LDS read -> row reduction by DPP ->
        // centre -> row reduction -> scale): with a branch per store the passes ran strictly one after the other.
        auto passes = [&](auto CH) {
            constexpr bool chained_c = decltype(CH)::value;
#pragma unroll
            for (int pass = 0; pass < LNP; ++pass) {
                const int row = (BMC / 8) * wave + 4 * pass + rl;
                const bool live = m0 + row < mend;
                f32x4 t[4];
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    t[v] = *reinterpret_cast<const f32x4 *>(hs + row * RS + 16 * (cl + 16 * v)) + xr[pass][v];      // (zeros without a residual)
                float *xrow = live ? p.x + (size_t)(m0 + row) * D + 4 * cl : p.dump + 4 * tid;
                const int xstep = live ? 64 : 0;
                if constexpr (TAPS) {
                    if (live && st.tap_pre)
#pragma unroll
                        for (int v = 0; v < 4; ++v) *reinterpret_cast<f32x4 *>(st.tap_pre + (size_t)(m0 + row) * D + 4 * cl + 64 * v) = t[v];
                }
                if constexpr (!chained_c) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) *reinterpret_cast<f32x4 *>(xrow + xstep * v) = t[v];
                    normalise(t, 0);
                } else {
                    normalise(t, 0);
#pragma unroll
                    for (int v = 0; v < 4; ++v) *reinterpret_cast<f32x4 *>(xrow + xstep * v) = t[v];
                    if constexpr (TAPS) {
                        if (live && st.tap_post)
#pragma unroll
                            for (int v = 0; v < 4; ++v) *reinterpret_cast<f32x4 *>(st.tap_post + (size_t)(m0 + row) * D + 4 * cl + 64 * v) = t[v];
                    }
                    normalise(t, 1);
                }
                const bool wxn = live && st.store_xn;
                T *nrow = wxn ? p.xn + (size_t)(m0 + row) * D + 4 * cl : reinterpret_cast<T *>(p.dump + 4 * tid);
                const int nstep = wxn ? 64 : 0;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const bf16x4 o = {(T)t[v][0], (T)t[v][1], (T)t[v][2], (T)t[v][3]};
                    *reinterpret_cast<bf16x4 *>(xa + v * PANEL + row * 128 + ((((cl >> 1) ^ (row & 7))) << 4) + (cl & 1) * 8) = o;
                    *reinterpret_cast<bf16x4 *>(nrow + nstep * v) = o;
                }
            }
        };
        if (chained) passes(std::true_type{}); else passes(std::false_type{});
        lds_fence_barrier();                                 // new operand image complete, staging consumed
    };
    // cooperative, coalesced copy of a staged bf16 [96][256] tile (row stride OS) to global through `store`
    auto flush_tile = [&](const unsigned char *tile, int col0, auto &&store) {
        for (int id = tid; id < BMC * 32; id += 512) {
            const int row = id >> 5, ch = id & 31;
            if (m0 + row < mend) store(m0 + row, col0 + ch * 8, reinterpret_cast<const T *>(tile + row * OS + ch * 16));
        }
    };

    auto run_stage = [&](auto KIND, const ChainStage &st, const T *after) {      // `after`: this wave's first slice of the next stage
        constexpr int kind = decltype(KIND)::value;
        if constexpr (kind == ST_ROWLN) {
            request_ln_params(st);
            const Bias2 bb = load_bias(st.bias);
            const Resid res = load_resid(st);
            f32x4 acc[MT][2];
            step(xa, acc, after, no_side, std::true_type{});
            stamp();
            lds_fence_barrier();                             // hs (hidden chunks / output tiles of the previous stage) is free
            stage_rows(acc, bb, st.alpha);
            stamp();
            rowln_epilogue(st, res);
            stamp();
        } else if constexpr (kind == ST_FFN) {
            // Software pipeline over the 256-wide hidden chunks:   P1(c): hidden(c) = xa W1(c)^T  (MFMA)
            //                                                      P2(c-1): out += silu(hidden(c-1)) W2(c-1)^T  (MFMA)  beside
            //                                                      S(c):  bias + SiLU of hidden(c) -> LDS        (VALU, folded into P2(c-1)'s k-steps)
            // weight stream order: W1(0), W1(1), W2(0), W1(2), W2(1), ..., W2(last).
            const int FF = st.N, nchunks = FF / 256;
            auto w1 = [&](int c) { return st.W + (size_t)(c * 8 + wave) * SLICE; };
            auto w2 = [&](int c) { return st.W2 + ((size_t)wave * (FF / 32) + c * 8) * 1024; };
            f32x4 acc1[MT][2], acc2[MT][2];
            Bias2 bb;
            auto silu_tile = [&](int tIdx, unsigned char *hb) {          // tile t = (row tile t / 2, column tile t % 2) of acc1 -> hb
                const int i = tIdx >> 1, j = tIdx & 1;
                const int jj = 32 * wave + 16 * j + 4 * g;               // hidden column inside the chunk
                const int ch16 = (jj & 63) >> 3, row = 16 * i + r16;
                // packed fp32 arithmetic around the two transcendentals (a SIMD issues VALU and MFMA instructions one at a time, so
                // this epilogue is paid in full beside the product step: 5 packed + 4 transcendental + 1 convert per value pair)
                typedef float f32x2_t __attribute__((ext_vector_type(2)));
                const f32x4 v4 = acc1[i][j] + bb.v[j];
                unsigned packed[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x2_t v = {v4[2 * h], v4[2 * h + 1]};
                    f32x2_t e = v * (f32x2_t){-1.44269504088896340736f, -1.44269504088896340736f};
                    e = (f32x2_t){__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])} + (f32x2_t){1.0f, 1.0f};
                    const f32x2_t o = v * (f32x2_t){__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
                    packed[h] = __builtin_bit_cast(unsigned, __builtin_convertvector(o, bf16x2));
                }
                typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                *reinterpret_cast<u32x2_t *>(hb + (jj >> 6) * PANEL + row * 128 + ((ch16 ^ (row & 7)) << 4) + ((jj & 7) >> 2) * 8) = (u32x2_t){packed[0], packed[1]};
            };
            request_ln_params(st);
            zero(acc2);
            bb = load_bias(st.bias);
            step(xa, acc1, nchunks > 1 ? w1(1) : w2(0), no_side, std::true_type{});    // P1(0)
            stamp();
#pragma unroll
            for (int t2 = 0; t2 < 2 * MT; ++t2) silu_tile(t2, hs);                        // S(0)
            stamp();
            lds_fence_barrier();
            stamp();
            for (int c = 1; c < nchunks; ++c) {
                bb = load_bias(st.bias + c * 256);
                step(xa, acc1, w2(c - 1), no_side, std::true_type{});                     // P1(c)
                stamp();
                unsigned char *hb = hs + (c & 1) * IMG;
                step(hs + ((c - 1) & 1) * IMG, acc2, c + 1 < nchunks ? w1(c + 1) : w2(c),    // P2(c-1) beside S(c)
                     [&](auto, int kk) {                                                  // the 2 MT tiles spread over the 8 k-steps
#pragma unroll
                         for (int t2 = 0; t2 < 2 * MT; ++t2) if ((t2 * 8) / (2 * MT) == kk) silu_tile(t2, hb);
                     }, std::false_type{});
                stamp();
                lds_fence_barrier();
                stamp();
            }
            const Bias2 b2 = load_bias(st.bias2);
            const Resid res = load_resid(st);                // (the hidden-chunk accumulators are dead: their registers carry the rows)
            step(hs + ((nchunks - 1) & 1) * IMG, acc2, after, no_side, std::false_type{});   // P2(last)
            stamp();
            lds_fence_barrier();                             // every wave is done reading the hidden chunks
            stage_rows(acc2, b2, st.alpha);
            stamp();
            rowln_epilogue(st, res);
            stamp();
        } else if constexpr (kind == ST_GLU) {
            // two steps of 256 packed columns: wave w's pair = (value tile, gate tile) of channels step*128 + 16w .. +15
            EpiGLU<T> e{st.out, D, st.bias, 2 * D};
            lds_fence_barrier();                             // hs free
#pragma unroll 1
            for (int s2 = 0; s2 < 2; ++s2) {
                const Bias2 bb = load_bias(st.bias + s2 * 256);          // v[0]: value bias, v[1]: gate bias
                f32x4 acc[MT][2];
                step(xa, acc, s2 == 0 ? st.W + (size_t)(8 + wave) * SLICE : after, no_side, std::true_type{});
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int row = 16 * i + r16;
                    bf16x4 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) o[q] = (T)((acc[i][0][q] + bb.v[0][q]) * sigmoid_f(acc[i][1][q] + bb.v[1][q]));
                    *reinterpret_cast<bf16x4 *>(hs + row * OS + (s2 * 128 + 16 * wave + 4 * g) * 2) = o;
                }
            }
            lds_fence_barrier();
            flush_tile(hs, 0, [&](int m, int c, const T *src) { e.store(m, c, src, 8); });
        } else if constexpr (kind == ST_QKV) {   // three steps of 256 columns
            lds_fence_barrier();                             // hs free
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) {
                const Bias2 bb = load_bias(st.bias + s3 * 256);
                f32x4 acc[MT][2];
                step(xa, acc, s3 < 2 ? st.W + (size_t)((s3 + 1) * 8 + wave) * SLICE : after, no_side, std::true_type{});
                stamp();
                unsigned char *tile = hs + (s3 & 1) * (BMC * OS);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const int row = 16 * i + r16;
                        bf16x4 o;
#pragma unroll
                        for (int q = 0; q < 4; ++q) o[q] = (T)(acc[i][j][q] + bb.v[j][q]);
                        *reinterpret_cast<bf16x4 *>(tile + row * OS + (32 * wave + 16 * j + 4 * g) * 2) = o;
                    }
                lds_fence_barrier();                         // tile s3 complete (tile s3-1 was flushed before this barrier)
                stamp();
                {   // step s3 is exactly q, k or v (256 = D columns each); thread -> one 8-column chunk of 6 rows
                    T *base = s3 == 0 ? st.q : (s3 == 1 ? st.k : st.v);
                    const int ch = tid & 31, hd = ch * 8, hh = hd / p.dh, d = hd - hh * p.dh;
                    base += (size_t)hh * p.Tp * p.dhp + d;
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const int row = (tid >> 5) + 16 * i;
                        if (m0 + row < mend) copy16(base + rowoff[row], reinterpret_cast<const T *>(tile + row * OS + ch * 16));
                    }
                }
                stamp();
            }
        }
    };
    auto first_slice = [&](int i) -> const T * { return p.st[i].W + (size_t)wave * SLICE; };
    run_stage(std::integral_constant<int, K0>{}, p.st[0], K1 >= 0 ? first_slice(1) : first_slice(0));
    run_stage(std::integral_constant<int, K1>{}, p.st[1], K2 >= 0 ? first_slice(2) : first_slice(0));
    run_stage(std::integral_constant<int, K2>{}, p.st[2], K3 >= 0 ? first_slice(3) : first_slice(0));
    run_stage(std::integral_constant<int, K3>{}, p.st[3], first_slice(0));
}

template <int MT, int DWK, int K0, int K1, int K2, int K3, bool TAPS>
static inline hipError_t launch_chain96_mt(hipStream_t s, const ChainArgs &a) {
    const size_t lds = (size_t)3 * 4 * (16 * MT) * 128 + 4096 + 4096 + 2048;      // operand image, 2 hidden images (+ slack), LayerNorm parameters, row offsets + frame indices
    auto kern = chain96_kernel<MT, DWK, K0, K1, K2, K3, TAPS>;
    hipError_t e = raise_lds_limit((const void *)kern, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(ceil_div(a.M, 16 * MT)), dim3(512), lds, s, a);
    return hipGetLastError();
}
// 96-row workgroups when they fill a useful share of the chip (M >= 4800 rows: 50 workgroups), 32-row workgroups below
template <int DWK, int K0, int K1, int K2, int K3>
static inline hipError_t launch_chain96_cfg(hipStream_t s, const ChainArgs &a, bool taps) {
    if (taps) return a.M >= COCR_CHAIN_SMALL_M ? launch_chain96_mt<6, DWK, K0, K1, K2, K3, true>(s, a) : launch_chain96_mt<2, DWK, K0, K1, K2, K3, true>(s, a);
    return a.M >= COCR_CHAIN_SMALL_M ? launch_chain96_mt<6, DWK, K0, K1, K2, K3, false>(s, a) : launch_chain96_mt<2, DWK, K0, K1, K2, K3, false>(s, a);
}

// the chain shapes the forward uses; stage weights point at the fragment-major copies
static inline hipError_t launch_chain96(hipStream_t s, const ChainArgs &a, bool taps = false) {
    const int k0 = a.st[0].kind, k1 = a.nstages > 1 ? a.st[1].kind : -1, k2 = a.nstages > 2 ? a.st[2].kind : -1, k3 = a.nstages > 3 ? a.st[3].kind : -1;
    if (a.dw_in) {                       // depthwise-conv prologue (kernel 31): the chains that follow the conv module's GLU
        if (k0 == ST_ROWLN && k1 == ST_FFN && k2 == ST_FFN && k3 == ST_QKV) return launch_chain96_cfg<31, ST_ROWLN, ST_FFN, ST_FFN, ST_QKV>(s, a, taps);
        if (k0 == ST_ROWLN && k1 == ST_FFN && k2 == -1) return launch_chain96_cfg<31, ST_ROWLN, ST_FFN, -1, -1>(s, a, taps);
        return hipErrorInvalidValue;
    }
    if (k0 == ST_FFN && k1 == ST_QKV && k2 == -1) return launch_chain96_cfg<0, ST_FFN, ST_QKV, -1, -1>(s, a, taps);
    if (k0 == ST_ROWLN && k1 == ST_GLU && k2 == -1) return launch_chain96_cfg<0, ST_ROWLN, ST_GLU, -1, -1>(s, a, taps);
    if (k0 == ST_ROWLN && k1 == ST_FFN && k2 == ST_FFN && k3 == ST_QKV) return launch_chain96_cfg<0, ST_ROWLN, ST_FFN, ST_FFN, ST_QKV>(s, a, taps);
    if (k0 == ST_ROWLN && k1 == ST_FFN && k2 == -1) return launch_chain96_cfg<0, ST_ROWLN, ST_FFN, -1, -1>(s, a, taps);
    return hipErrorInvalidValue;
}
