// Fused feed-forward module (reference conformer/feed_forward.py:45-52 inside ResidualConnectionModule,
// modules.py:32, encoder.py:62-75):   x <- x + f * (W2 silu(W1 xn + b1) + b2),   xn = LayerNorm(x) given,
// followed in the same kernel by the LayerNorm(s) the reference applies next (EpiResidualLN, gemm.hip.h).
//
// One workgroup owns 64 rows for the whole module; the 4D-wide hidden activation never leaves the CU:
//   per hidden chunk of 128 columns:   H = silu(Xn W1[chunk]^T + b1[chunk])      64 x 128, fp32 acc -> bf16 in LDS
//                                      Y += H W2[:, chunk]^T                      64 x D,   fp32 acc in registers
// The 64 x D operand tile is LDS resident; W1 / W2 are streamed exactly once per workgroup through a 3-deep
// LDS-DMA ring (global_load_lds, counted vmcnt, one raw barrier per tile; same swizzled lane-linear tile image as
// gemm_ring_kernel).  Per workgroup: 2 * 64 * D * 4D * 2 flop against D * 4D * 2 * 2 bytes of weights (L2 hits):
// 64 flop per streamed byte at D = 256.  bf16 operands only (the fp32 mode keeps the two-GEMM path).
#pragma once
#include <type_traits>

#include "gemm.hip.h"

template <int D, typename Epi, int MODE = 0>     // D = encoder_dim == 256; MODE 1: DMA only, 2: compute only (measurement builds)
__global__ __launch_bounds__(512) void ffn_fused_kernel(const bf16_t *__restrict__ xn, const bf16_t *__restrict__ W1, const float *__restrict__ b1,
                                                        const bf16_t *__restrict__ W2, int M, int FF, Epi epi) {
    typedef bf16_t T;
    static_assert(D == 256, "wave tiling below is written for encoder_dim 256");
    // 8 waves = 2 (rows) x 4 (columns): two waves per SIMD, so one wave's MFMAs overlap the other's LDS reads / VALU / waits
    // (with 4 waves -- one per SIMD -- the same loop measured compute-only 43 us against 23 us for its DMA stream alone).
    constexpr int DK = D / 64;                 // k-tiles of the first product = 128-byte panels of the operand tile
    constexpr int PANEL = 64 * 128;            // one [64 rows][128 B] panel
    constexpr int SLOT = 128 * 128;            // every streamed tile is 128 rows x 128 B = 16 KiB
    constexpr int NST = 6;                     // ring: 3 pair slots of 2 tiles; 2 pairs (64 KiB) in flight (1 block per CU: the ring IS the latency hiding)
    constexpr int PER = 2;                     // DMA wave-instructions per wave per tile (16 per tile / 8 waves)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *xs = smem;                               // DK panels
    unsigned char *hs = xs + DK * PANEL;                    // 2 panels
    unsigned char *ring = hs + 2 * PANEL;                   // NST slots
    float *b1s = reinterpret_cast<float *>(ring + NST * SLOT);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4, swz = r16 & 7;
    const int wm = wave >> 2, wn = wave & 3;
    const int m0 = blockIdx.x * 64;
    const int nchunks = FF / 128;
    const int lrow = lane >> 3, cpos = lane & 7;

    // residual rows of this wave's share of the LayerNorm epilogue: requested first, consumed last
    typename Epi::Rows4 xr[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) xr[it] = epi.rows4_load(m0 + wave * 4 + 32 * it, M, lane);
    // b1 -> LDS by DMA as well (FF * 4 bytes = FF / 256 wave-instructions of 1 KiB)
    for (int i = wave; i < FF / 256; i += 8)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(b1 + i * 256 + lane * 4), (lds_ptr_t)(reinterpret_cast<unsigned char *>(b1s) + i * 1024), 16, 0, 0);

    // operand tile: DK panels x 8 row groups = 32 wave-instructions, rows clamped at M-1
#pragma unroll
    for (int i = 0; i < DK; ++i) {
        const int id = wave + 8 * i, pnl = id >> 3, rg = id & 7, row = rg * 8 + lrow;
        const T *src = xn + (size_t)min(m0 + row, M - 1) * D + pnl * 64 + ((cpos ^ (row & 7)) * 8);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(xs + pnl * PANEL + rg * 1024), 16, 0, 0);
    }
    // The weight stream is walked in PAIRS of 16 KiB tiles (one barrier, one wait per pair): per hidden chunk
    //   pair 0: W1 k-tiles 0,1   pair 1: W1 k-tiles 2,3 (+ hidden chunk written)   pair 2: W2 k-half 0, column halves 0,1   pair 3: W2 k-half 1
    // The pair type is a compile-time property of its position (the accumulators of the second product are then statically
    // indexed: a runtime selector made the compiler shuffle every accumulator register every tile).  Ring = 3 pair slots.
    auto issue_pair = [&](int c, auto KP) {
        constexpr int kp = decltype(KP)::value;
        unsigned char *st = ring + (((c * 4 + kp) % 3) * 2) * SLOT;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int rg = wave + 8 * i, row = rg * 8 + lrow;
                const T *src;
                if constexpr (kp < 2) src = W1 + ((size_t)c * 128 + row) * D + (2 * kp + h) * 64 + ((cpos ^ (row & 7)) * 8);        // W1 rows c*128 + row
                else src = W2 + (size_t)(h * 128 + row) * FF + c * 128 + (kp - 2) * 64 + ((cpos ^ (row & 7)) * 8);                    // W2 rows = output columns h*128 + row
                if constexpr (MODE != 2) __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(st + h * SLOT + rg * 1024), 16, 0, 0);
            }
    };
    issue_pair(0, std::integral_constant<int, 0>{});
    issue_pair(0, std::integral_constant<int, 1>{});

    f32x4 acc1[2][2], acc2[2][4];                           // wave tile 32 x 32 of the hidden chunk; 32 rows x (2 halves x 32 columns) of the output
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int npairs = nchunks * 4;
    // The wave's 32 operand rows stay in registers for the whole kernel (2 x 8 fragments = 64 VGPRs): the first product then
    // reads only weight fragments from LDS.  The operand DMAs were issued before the two weight pairs (4 * PER instructions).
    wait_vmcnt<4 * PER>();
    __builtin_amdgcn_s_barrier();
    bf16x8 xa[2][2 * DK];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kk = 0; kk < 2 * DK; ++kk)
            xa[i][kk] = lds_frag_swz(xs + (kk >> 1) * PANEL + (wm * 32 + 16 * i + r16) * 128, kk & 1, g, swz, T());

    auto step = [&](int c, auto KP) {
        constexpr int kp = decltype(KP)::value;
        const int P = c * 4 + kp;
        // pair P landed: at this point pairs 0 .. P+1 have been issued, so only pair P+1 (2 * PER instructions) may be in flight
        if (P + 1 < npairs) wait_vmcnt<2 * PER>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        {   // refill the pair slot every wave finished reading before this barrier: pair P + 2
            constexpr int nkp = (kp + 2) & 3, cadd = (kp + 2) >> 2;
            if (c + cadd < nchunks) issue_pair(c + cadd, std::integral_constant<int, nkp>{});
        }
        if constexpr (MODE == 1) return;
        const unsigned char *st = ring + ((P % 3) * 2) * SLOT;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned char *sw = st + h * SLOT + (wn * 32 + r16) * 128;
            if constexpr (kp < 2) {
                constexpr int kt = 2 * kp;
                if (kp == 0 && h == 0) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    bf16x8 b[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) b[j] = lds_frag_swz(sw + j * 16 * 128, kc, g, swz, T());
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc1[i][j] = mma16(b[j], xa[i][2 * (kt + h) + kc], acc1[i][j]);
                }
            } else {
                const unsigned char *sa = hs + (kp - 2) * PANEL + (wm * 32 + r16) * 128;
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    bf16x8 a[2], b[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) a[i] = lds_frag_swz(sa + i * 16 * 128, kc, g, swz, T());
#pragma unroll
                    for (int j = 0; j < 2; ++j) b[j] = lds_frag_swz(sw + j * 16 * 128, kc, g, swz, T());
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc2[i][2 * h + j] = mma16(b[j], a[i], acc2[i][2 * h + j]);
                }
            }
        }
        if constexpr (kp == 1) {      // hidden chunk: bias + SiLU, bf16, into the swizzled operand image of the second product
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wm * 32 + 16 * i + r16;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int jj = wn * 32 + 16 * j + 4 * g;                 // hidden column inside the chunk
                    const f32x4 bb = *reinterpret_cast<const f32x4 *>(b1s + c * 128 + jj);
                    bf16x4 hv;
#pragma unroll
                    for (int q = 0; q < 4; ++q) hv[q] = (T)silu_f(acc1[i][j][q] + bb[q]);
                    const int ch16 = (jj & 63) >> 3;
                    *reinterpret_cast<bf16x4 *>(hs + (jj >> 6) * PANEL + row * 128 + ((ch16 ^ (row & 7)) << 4) + ((jj & 7) >> 2) * 8) = hv;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // visible to the other waves at the next barrier
        }
    };
    for (int c = 0; c < nchunks; ++c) {
        step(c, std::integral_constant<int, 0>{}); step(c, std::integral_constant<int, 1>{});
        step(c, std::integral_constant<int, 2>{}); step(c, std::integral_constant<int, 3>{});
    }
    __syncthreads();
    // ---- epilogue: alpha (acc + b2) staged as fp32 rows in the (now idle) ring, then residual + LayerNorm per row
    constexpr int RS = D * 4 + 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wm * 32 + 16 * i + r16;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int n = (jj >> 1) * 128 + wn * 32 + (jj & 1) * 16 + 4 * g;
            const float v[4] = {acc2[i][jj][0], acc2[i][jj][1], acc2[i][jj][2], acc2[i][jj][3]};
            float r[4];
            epi.transform(n, v, r);
            *reinterpret_cast<f32x4 *>(ring + row * RS + n * 4) = (f32x4){r[0], r[1], r[2], r[3]};
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int rr = wave * 4 + 32 * it;
        epi.rows4(m0 + rr, M, reinterpret_cast<const float *>(ring + rr * RS), RS / 4, lane, xr[it]);
    }
}

template <typename Epi> static inline bool ffn_fused_supported(int D, int FF) { return D == 256 && FF % 256 == 0 && FF >= 256; }

template <typename Epi, int MODE = 0>
static inline hipError_t launch_ffn_fused(hipStream_t s, const bf16_t *xn, const bf16_t *W1, const float *b1, const bf16_t *W2, int M, int D, int FF,
                                          const Epi &epi) {
    constexpr int DD = 256;
    const size_t lds = (size_t)(DD / 64) * 8192 + 2 * 8192 + 6 * 16384 + (size_t)FF * 4;
    auto kern = ffn_fused_kernel<DD, Epi, MODE>;
    hipError_t e = raise_lds_limit((const void *)kern, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(ceil_div(M, 64)), dim3(512), lds, s, xn, W1, b1, W2, M, FF, epi);
    return hipGetLastError();
}
