// C[M,N] = A[M,K] . W[N,K]^T with fused epilogues -- every dense contraction of the path:
// frontend pointwise conv F3 and flatten-linear F5 (reference conformer/convolution.py:207-213,224),
// FFN linears (feed_forward.py:47,50), q/k/v/out projections (attention.py:59-61,70), the conv
// module's pointwise convs (convolution.py:138,143), the positional projection (attention.py:62) and
// the decoder (pred.py:90).  Both operands are k-contiguous, which is the MFMA fragment order
// (common.hip.h), so activations (M,K) and nn.Linear weights (N,K) are used as stored.
//
// Tile BMxBN per 256-thread workgroup (4 waves as 2x2), k-tile = 128 bytes per row, operands staged
// global -> registers -> LDS (rows padded by 16 B against ds_read_b128 bank conflicts), two LDS
// buffers, one barrier per k-tile: the next tile's global loads are in flight during the MFMAs.
#pragma once
#include "common.hip.h"

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2 };

// ---- epilogues: called once per valid output element (m < M, n < N) with the fp32 accumulator ----

// out[m, n] = act(acc + bias[n]) stored as T (hidden activations)
template <typename T, int ACT> struct EpiBiasAct {
    static constexpr bool PAIRED = false;
    T *out; int ldo; const float *bias;
    __device__ __forceinline__ void operator()(int m, int n, float v) const {
        v += bias[n];
        if (ACT == ACT_RELU) v = fmaxf(v, 0.0f);
        if (ACT == ACT_SILU) v = silu_f(v);
        out[(size_t)m * ldo + n] = from_f32<T>(v);
    }
};

// out[m, n] = acc (+ bias[n]) as fp32 (frontend output = residual stream; decoder logits)
struct EpiStoreF32 {
    static constexpr bool PAIRED = false;
    float *out; int ldo; const float *bias;
    __device__ __forceinline__ void operator()(int m, int n, float v) const {
        out[(size_t)m * ldo + n] = bias ? v + bias[n] : v;
    }
};

// x[m, n] += alpha * (acc + bias[n])  -- ResidualConnectionModule, modules.py:32 (input_factor 1)
struct EpiResidual {
    static constexpr bool PAIRED = false;
    float *x; int ldx; const float *bias; float alpha;
    __device__ __forceinline__ void operator()(int m, int n, float v) const {
        float *p = x + (size_t)m * ldx + n;
        *p = *p + alpha * (v + bias[n]);
    }
};

// GLU over channels (convolution.py:139): the weight rows are interleaved at load time so that the
// 16-column tile 2j holds the value half of channels 16j..16j+15 and tile 2j+1 their gates;
// out[m, 16j + c] = (a + ba) * sigmoid(g + bg)
template <typename T> struct EpiGLU {
    static constexpr bool PAIRED = true;
    T *out; int ldo; const float *bias;
    __device__ __forceinline__ void operator()(int m, int n, float a, float g) const {
        a += bias[n];
        g += bias[n + 16];
        const int ch = ((n >> 5) << 4) + (n & 15);
        out[(size_t)m * ldo + ch] = from_f32<T>(a * sigmoid_f(g));
    }
};

// fused q/k/v projection (N = 3D): scatter into the attention layouts
//   q, k : [B][h][Tp][dhp]   (head dim padded to a multiple of 32 with zeros, rows >= T unused)
//   vt   : [B][h][dhp][Tp]   (V transposed: keys contiguous, the B-operand order of P.V)
template <typename T> struct EpiQKV {
    static constexpr bool PAIRED = false;
    T *q, *k, *vt; const float *bias; int D, dh, dhp, heads, T_, Tp;
    __device__ __forceinline__ void operator()(int m, int n, float v) const {
        v += bias[n];
        const int which = n / D, hd = n - which * D;
        const int hh = hd / dh, d = hd - hh * dh;
        const int b = m / T_, t = m - b * T_;
        const size_t bh = (size_t)b * heads + hh;
        if (which == 2) vt[(bh * dhp + d) * Tp + t] = from_f32<T>(v);
        else (which == 0 ? q : k)[(bh * Tp + t) * dhp + d] = from_f32<T>(v);
    }
};

// positional projection table P = PE Wpos^T into [row][h][dhp] (head-padded)
template <typename T> struct EpiPosTable {
    static constexpr bool PAIRED = false;
    T *out; int dh, dhp, heads;
    __device__ __forceinline__ void operator()(int m, int n, float v) const {
        const int hh = n / dh, d = n - hh * dh;
        out[((size_t)m * heads + hh) * dhp + d] = from_f32<T>(v);
    }
};

// ---- kernel -----------------------------------------------------------------------------------

template <typename T, int BM, int BN, typename Epi>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const T *__restrict__ A, int lda, const T *__restrict__ W, int ldw,
                                                      int M, int N, int K, Epi epi) {
    constexpr int ROWB = 128;               // bytes of k per row per tile
    constexpr int BK = ROWB / sizeof(T);    // bf16: 64, fp32: 32
    constexpr int EPC = 16 / sizeof(T);     // elements per 16-byte chunk
    constexpr int STRIDE = ROWB + 16;       // padded LDS row
    constexpr int MI = BM / 32, NI = BN / 32;
    constexpr int A_IT = BM * 8 / 256, W_IT = BN * 8 / 256;
    static_assert(BM % 32 == 0 && BN % 32 == 0, "tile");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][(BM + BN) * STRIDE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    uint4 ra[A_IT], rw[W_IT];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int id = it * 256 + tid, row = id >> 3, ch = id & 7;
            const int m = m0 + row, k = k0 + ch * EPC;
            ra[it] = (m < M && k < K) ? *reinterpret_cast<const uint4 *>(A + (size_t)m * lda + k) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int id = it * 256 + tid, row = id >> 3, ch = id & 7;
            const int n = n0 + row, k = k0 + ch * EPC;
            rw[it] = (n < N && k < K) ? *reinterpret_cast<const uint4 *>(W + (size_t)n * ldw + k) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int id = it * 256 + tid, row = id >> 3, ch = id & 7;
            *reinterpret_cast<uint4 *>(&smem[buf][row * STRIDE + ch * 16]) = ra[it];
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int id = it * 256 + tid, row = id >> 3, ch = id & 7;
            *reinterpret_cast<uint4 *>(&smem[buf][(BM + row) * STRIDE + ch * 16]) = rw[it];
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
        const unsigned char *sa = &smem[buf][(wm * (BM / 2) + r16) * STRIDE];
        const unsigned char *sw = &smem[buf][(BM + wn * (BN / 2) + r16) * STRIDE];
#pragma unroll
        for (int kc = 0; kc < BK / 32; ++kc) {
            const int off = kc * 32 * (int)sizeof(T) + g * 8 * (int)sizeof(T);
            typename FragOf<T>::type a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = load_frag(reinterpret_cast<const T *>(sa + i * 16 * STRIDE + off));
#pragma unroll
            for (int j = 0; j < NI; ++j) b[j] = load_frag(reinterpret_cast<const T *>(sw + j * 16 * STRIDE + off));
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = mma16(a[i], b[j], acc[i][j]);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // epilogue: accumulator (i, j), reg q is element (row 4g + q, column r16) of its 16x16 tile
    const int mb = m0 + wm * (BM / 2) + 4 * g, nb = n0 + wn * (BN / 2) + r16;
    if constexpr (Epi::PAIRED) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; j += 2) {
                const int n = nb + j * 16;
                if (n + 16 < N)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int m = mb + i * 16 + q;
                        if (m < M) epi(m, n, acc[i][j][q], acc[i][j + 1][q]);
                    }
            }
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int n = nb + j * 16;
                if (n < N)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int m = mb + i * 16 + q;
                        if (m < M) epi(m, n, acc[i][j][q]);
                    }
            }
    }
}

// Host launcher.  Requirements (checked by the caller at model build): K % (16/sizeof(T)) == 0,
// lda/ldw multiples of 16 bytes, base pointers 16-byte aligned.
template <typename T, typename Epi>
static inline void launch_gemm(hipStream_t s, const T *A, int lda, const T *W, int ldw, int M, int N, int K, const Epi &epi) {
    // small-N products at M ~ 10^4 would leave most of the 256 CUs idle with 128x128 tiles
    const long blocks128 = (long)ceil_div(M, 128) * ceil_div(N, 128);
    if (blocks128 >= 512) {
        dim3 grid(ceil_div(N, 128), ceil_div(M, 128));
        hipLaunchKernelGGL((gemm_nt_kernel<T, 128, 128, Epi>), grid, dim3(256), 0, s, A, lda, W, ldw, M, N, K, epi);
    } else {
        dim3 grid(ceil_div(N, 64), ceil_div(M, 64));
        hipLaunchKernelGGL((gemm_nt_kernel<T, 64, 64, Epi>), grid, dim3(256), 0, s, A, lda, W, ldw, M, N, K, epi);
    }
}
