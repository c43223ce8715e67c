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
};

struct ChainArgs {
    const bf16_t *A0;      // first operand rows (M, 256)
    float *x;              // fp32 residual stream (M, 256)
    bf16_t *xn;            // normalised operand (M, 256), written when a stage asks for it
    int M, nstages;
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
