// Fused feed-forward module (reference conformer/feed_forward.py:45-52 inside ResidualConnectionModule,
// modules.py:32, encoder.py:62-75):   x <- x + f * (W2 silu(W1 xn + b1) + b2),   xn = LayerNorm(x) given,
// followed in the same kernel by the LayerNorm(s) the reference applies next (EpiResidualLN, gemm.hip.h).
//
// One workgroup owns 48 rows for the whole module; the 4D-wide hidden activation never leaves the CU:
//   per hidden chunk of 128 columns:   H = silu(Xn W1[chunk]^T + b1[chunk])      48 x 128, fp32 acc -> bf16 in LDS
//                                      Y += H W2[:, chunk]^T                      48 x D,   fp32 acc in registers
// W1 / W2 are streamed exactly once per workgroup by LDS-DMA (global_load_lds, counted vmcnt, the swizzled lane-linear
// tile image of gemm_ring_kernel).  Per workgroup: 2 * 48 * D * 4D * 2 flop against D * 4D * 2 * 2 bytes of weights
// (L2 hits): 48 flop per streamed byte at D = 256.  bf16 operands only (the fp32 mode keeps the two-GEMM path).
#pragma once
#include <type_traits>

#include "gemm.hip.h"

template <int D, typename Epi>     // D = encoder_dim == 256
__global__ __launch_bounds__(512) void ffn_fused_kernel(const bf16_t *__restrict__ xn, const bf16_t *__restrict__ W1, const float *__restrict__ b1,
                                                        const bf16_t *__restrict__ W2, int M, int FF, Epi epi) {
    typedef bf16_t T;
    static_assert(D == 256, "wave tiling below is written for encoder_dim 256");
    // Workgroup = 48 rows (M = 9600 -> 200 workgroups = one per CU in a single round), 8 waves side by side (1 x 8).
    // Wave w owns hidden columns [16w, 16w+16) of every 128-wide hidden chunk and output columns [32w, 32w+32):
    //   * its weight rows are its own: W1 / W2 slices stream through WAVE-PRIVATE LDS buffers with the wave's own counted
    //     vmcnt -- no workgroup barrier guards the weight stream, waves drift apart and overlap each other's waits;
    //   * its 48 x 256 operand rows stay in registers (96 VGPRs) for the whole kernel;
    //   * only the hidden chunk (48 x 128 bf16, double buffered) crosses waves: ONE barrier per chunk.
    // Per chunk and wave: 48 MFMAs against 28 LDS fragment reads and 16 DMA wave-instructions (16 KiB of weights).
    constexpr int BMC = 48;
    constexpr int KC1 = D / 32;                 // 8 k-chunks of the first product
    constexpr int HPANEL = BMC * 128;           // one [48 rows][128 B] panel of the hidden chunk / operand tile
    constexpr int WBUF = 8192;                  // per wave: W1 slice [4 panels][16 rows][128 B]; W2 slice [2 panels][32 rows][128 B]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *hs = smem;                               // 2 buffers x 2 panels (= 4 panels: holds the operand tile during the prologue)
    unsigned char *wreg = hs + 4 * HPANEL;                  // 8 waves x (W1 slice + W2 slice)
    float *b1s = reinterpret_cast<float *>(wreg + 8 * 2 * WBUF);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4, swz = r16 & 7;
    const int m0 = blockIdx.x * BMC, mend = min(M, m0 + BMC);
    const int nchunks = FF / 128;
    const int lrow = lane >> 3, cpos = lane & 7;
    unsigned char *w1b = wreg + wave * 2 * WBUF, *w2b = w1b + WBUF;

    // residual rows of this wave's share of the LayerNorm epilogue (row groups wave and wave + 8 of 12): requested first, consumed last
    typename Epi::Rows4 xr[2];
    xr[0] = epi.rows4_load(m0 + 4 * wave, mend, lane);
    xr[1] = epi.rows4_load(m0 + 4 * min(wave + 8, 11), mend, lane);
    // b1 -> LDS by DMA (FF * 4 bytes = FF / 256 wave-instructions of 1 KiB)
    for (int i = wave; i < FF / 256; i += 8)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(b1 + i * 256 + lane * 4), (lds_ptr_t)(reinterpret_cast<unsigned char *>(b1s) + i * 1024), 16, 0, 0);
    // operand tile: 4 panels x 6 row groups = 24 wave-instructions (3 per wave), rows clamped at M-1, into the hidden-chunk area
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int id = wave + 8 * i, pnl = id / 6, rg = id - pnl * 6, row = rg * 8 + lrow;
        const T *src = xn + (size_t)min(m0 + row, M - 1) * D + pnl * 64 + ((cpos ^ (row & 7)) * 8);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(hs + pnl * HPANEL + rg * 1024), 16, 0, 0);
    }
    auto issue_w1 = [&](int c) {      // rows c*128 + 16*wave + (0..15), all k: 4 panels x 2 wave-instructions
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pnl = i >> 1, row = (i & 1) * 8 + lrow;
            const T *src = W1 + ((size_t)c * 128 + 16 * wave + row) * D + pnl * 64 + ((cpos ^ (row & 7)) * 8);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(w1b + pnl * 2048 + (i & 1) * 1024), 16, 0, 0);
        }
    };
    auto issue_w2 = [&](int c) {      // rows (= output columns) 32*wave + (0..31), k = c*128 .. +127: 2 panels x 4 wave-instructions
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pnl = i >> 2, row = (i & 3) * 8 + lrow;
            const T *src = W2 + ((size_t)32 * wave + row) * FF + c * 128 + pnl * 64 + ((cpos ^ (row & 7)) * 8);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(w2b + pnl * 4096 + (i & 3) * 1024), 16, 0, 0);
        }
    };
    issue_w1(0);
    issue_w2(0);
    // operand rows -> registers (the 16 weight instructions above may stay in flight), then the area is free for hidden chunks
    wait_vmcnt<16>();
    __builtin_amdgcn_s_barrier();
    bf16x8 xa[3][KC1];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int kk = 0; kk < KC1; ++kk) xa[i][kk] = lds_frag_swz(hs + (kk >> 1) * HPANEL + (16 * i + r16) * 128, kk & 1, g, swz, T());
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    f32x4 acc2[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int c = 0; c < nchunks; ++c) {
        // ---- first product: hidden[48][16w .. 16w+15] of chunk c;  W1(c) landed (only W2(c) may be in flight)
        wait_vmcnt<8>();
        f32x4 acc1[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) acc1[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KC1; ++kk) {
            const bf16x8 b = lds_frag_swz(w1b + (kk >> 1) * 2048 + r16 * 128, kk & 1, g, swz, T());
#pragma unroll
            for (int i = 0; i < 3; ++i) acc1[i] = mma16(b, xa[i][kk], acc1[i]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // own reads of the W1 slice retired before it is refilled
        if (c + 1 < nchunks) issue_w1(c + 1);
        {   // bias + SiLU, bf16, into the swizzled operand image of the second product (buffer c & 1)
            unsigned char *hb = hs + (c & 1) * 2 * HPANEL;
            const int jj = 16 * wave + 4 * g;                    // hidden column inside the chunk
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
        __builtin_amdgcn_s_barrier();                            // the whole hidden chunk is in LDS
        // ---- second product: out[48][32w .. 32w+31] += hidden_chunk . W2[:, chunk]^T;  W2(c) landed (only W1(c+1) may be in flight)
        if (c + 1 < nchunks) wait_vmcnt<8>(); else wait_vmcnt<0>();
        const unsigned char *hb = hs + (c & 1) * 2 * HPANEL;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 a[3], b[2];
#pragma unroll
            for (int i = 0; i < 3; ++i) a[i] = lds_frag_swz(hb + (kk >> 1) * HPANEL + (16 * i + r16) * 128, kk & 1, g, swz, T());
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = lds_frag_swz(w2b + (kk >> 1) * 4096 + (16 * j + r16) * 128, kk & 1, g, swz, T());
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc2[i][j] = mma16(b[j], a[i], acc2[i][j]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // own reads of the W2 slice retired before it is refilled
        if (c + 1 < nchunks) issue_w2(c + 1);
    }
    __syncthreads();
    // ---- epilogue: alpha (acc + b2) staged as fp32 rows over the (now idle) weight buffers, then residual + LayerNorm per row
    constexpr int RS = D * 4 + 16;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = 16 * i + r16, n = 32 * wave + 16 * j + 4 * g;
            const float v[4] = {acc2[i][j][0], acc2[i][j][1], acc2[i][j][2], acc2[i][j][3]};
            float r[4];
            epi.transform(n, v, r);
            *reinterpret_cast<f32x4 *>(wreg + row * RS + n * 4) = (f32x4){r[0], r[1], r[2], r[3]};
        }
    __syncthreads();
    epi.rows4(m0 + 4 * wave, mend, reinterpret_cast<const float *>(wreg + 4 * wave * RS), RS / 4, lane, xr[0]);
    if (wave < 4) epi.rows4(m0 + 4 * (wave + 8), mend, reinterpret_cast<const float *>(wreg + 4 * (wave + 8) * RS), RS / 4, lane, xr[1]);
}

template <typename Epi> static inline bool ffn_fused_supported(int D, int FF) { return D == 256 && FF % 256 == 0 && FF >= 256; }

template <typename Epi>
static inline hipError_t launch_ffn_fused(hipStream_t s, const bf16_t *xn, const bf16_t *W1, const float *b1, const bf16_t *W2, int M, int D, int FF,
                                          const Epi &epi) {
    constexpr int DD = 256;
    const size_t lds = (size_t)4 * 48 * 128 + 8 * 2 * 8192 + (size_t)FF * 4;     // hidden chunks + per-wave weight slices + b1
    auto kern = ffn_fused_kernel<DD, Epi>;
    hipError_t e = raise_lds_limit((const void *)kern, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(ceil_div(M, 48)), dim3(512), lds, s, xn, W1, b1, W2, M, FF, epi);
    return hipGetLastError();
}
