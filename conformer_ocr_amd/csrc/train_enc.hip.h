// Training step of the whole network -- the kernels of `RecognitionModel.training_step` (reference model.py:129-152: forward in train
// mode, CTC criterion, `loss.backward()` through decoder and encoder) that are not matrix products.  fp32 throughout (the reference trains
// in fp32 unless told otherwise, cli/__init__.py:39-70): correctness-first kernels, one thread / one wave per output with fixed-order
// reductions (no floating-point atomics: two runs of a step give the same bits).  The matrix products -- forward, input gradient and
// weight gradient of every Linear / pointwise conv -- run on the exact-fp32 MFMA GEMM of gemm.hip.h through explicit transposes
// (train_api.hip.h).  Train-mode semantics: BatchNorm1d with batch statistics over ALL (line, frame) positions of the padded batch
// (convolution.py:141; no masking anywhere) and running-statistics update (momentum 0.1, unbiased variance); dropout at the
// reference's six sites (convolution.py:144,225; feed_forward.py:49,51; attention.py:98,151) from a counter-based generator, so
// the backward regenerates the masks instead of storing them.
#pragma once
#include "common.hip.h"

// ---- dropout: keep(seed, site, index) -------------------------------------------------------------------------------------------
__device__ __forceinline__ bool drop_keep(unsigned long long seed, unsigned site, unsigned long long idx, float p) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (idx + 1) + ((unsigned long long)site << 48);      // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(unsigned)(z >> 40) * (1.0f / 16777216.0f) >= p;
}
// x <- keep ? x / (1 - p) : 0 (forward on activations, backward on their gradients: the same mask)
__global__ static void k_dropout(float *x, size_t n, float p, unsigned long long seed, unsigned site) {
    const float sc = 1.0f / (1.0f - p);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        x[i] = drop_keep(seed, site, i, p) ? x[i] * sc : 0.f;
}

// Round 4: the dropout folded into the elementwise kernel beside it (the same values: the same multiply by 1 / (1 - p) at the same place in the
// chain; p = 0 leaves the value alone) -- a training step had 146 dropout launches, 72 of them behind a copy or a scaled copy.
__device__ __forceinline__ float drop_apply(float v, unsigned long long seed, unsigned site, unsigned long long idx, float p, float sc) {
    return p > 0.f ? (drop_keep(seed, site, idx, p) ? v * sc : 0.f) : v;
}
// y = a + alpha drop(b)          (a sublayer's residual add)
__global__ static void k_add3_drop(float *y, const float *a, const float *b, float alpha, size_t n, float p, unsigned long long seed, unsigned site) {
    const float sc = 1.0f / (1.0f - p);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = fmaf(alpha, drop_apply(b[i], seed, site, i, p, sc), a[i]);
}
// out = drop(alpha in)           (the gradient entering a sublayer's last dropout: zero + alpha x, dropped)
__global__ static void k_scale_drop(float *out, const float *in, float alpha, size_t n, float p, unsigned long long seed, unsigned site) {
    const float sc = 1.0f / (1.0f - p);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = drop_apply(fmaf(alpha, in[i], 0.f), seed, site, i, p, sc);
}
// ---- small elementwise helpers ------------------------------------------------------------------------------------------------------
__global__ static void k_axpy(float *y, const float *x, float alpha, size_t n) {          // y += alpha x
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = fmaf(alpha, x[i], y[i]);
}
__global__ static void k_add3(float *y, const float *a, const float *b, float alpha, size_t n) {      // y = a + alpha b
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = fmaf(alpha, b[i], a[i]);
}
__global__ static void k_u8_to_f32(const uint8_t *in, float *out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (float)in[i] * (1.0f / 255.0f);
}
// fp32 -> bf16 (round to nearest even), n a multiple of 4: the operands of the 'medium' matmul precision
__global__ static void k_f32_to_bf16(const float *__restrict__ in, bf16_t *__restrict__ out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = reinterpret_cast<const f32x4 *>(in)[i];
        reinterpret_cast<bf16x4 *>(out)[i] = (bf16x4){(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    }
}
// out[c * ldo + r] = in[r * C + c] for r < R, 0 for R <= r < ldo (grid.y covers ldo: no separate clearing of the tail)
__global__ static void k_transpose(const float *__restrict__ in, float *__restrict__ out, int R, int C, int ldo) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) tile[j][tx] = (r0 + j < R && c0 + tx < C) ? in[(size_t)(r0 + j) * C + c0 + tx] : 0.f;
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < C && r0 + tx < ldo) out[(size_t)(c0 + j) * ldo + r0 + tx] = tile[tx][j];
}
// 'medium' matmul precision (round 4): ONE pass over an fp32 (R, C) matrix for everything its weight-gradient and input-gradient products
// need -- outT (C, ldo) = its transpose in bf16, zero beyond row R (the operand of dW = dY^T X; the separate k_transpose + k_f32_to_bf16
// wrote and re-read the fp32 transpose); outR (R, C) = its bf16 copy (the operand of dX = dY W), or null; part[tile row][C] = the column sums
// of each 32-row tile, rows added in order (the bias gradient's partial sums: k_colsum_final adds the tiles in a fixed order), or null.
__global__ __launch_bounds__(256) static void k_transpose_bf16(const float *__restrict__ in, bf16_t *__restrict__ outT, bf16_t *__restrict__ outR,
                                                              float *__restrict__ part, int R, int C, int ldo) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const bool ok = r0 + j < R && c0 + tx < C;
        const float v = ok ? in[(size_t)(r0 + j) * C + c0 + tx] : 0.f;
        tile[j][tx] = v;
        if (outR && ok) outR[(size_t)(r0 + j) * C + c0 + tx] = (bf16_t)v;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < C && r0 + tx < ldo) outT[(size_t)(c0 + j) * ldo + r0 + tx] = (bf16_t)tile[tx][j];
    if (part && ty == 0 && c0 + tx < C && r0 < R) {
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) sum += tile[j][tx];
        part[(size_t)blockIdx.y * C + c0 + tx] = sum;
    }
}
// 'medium' matmul precision: the bf16 row-major copy of an fp32 (R, C) matrix with its rows padded by zeros to Rp (the K-major operand of the
// weight-gradient product gemm_tn_kernel, whose depth is a multiple of 32 x the split count), and -- part != null -- the column sums of every
// 32-row tile in a fixed order (the bias gradient's partial sums).  C % 4 == 0.  A thread = four columns x every fourth row of a 32-row tile.
__global__ __launch_bounds__(256) static void k_rows_bf16(const float *__restrict__ in, bf16_t *__restrict__ out, float *__restrict__ part, int R, int C, int Rp) {
    __shared__ f32x4 red4[4][64];
    const int cq = threadIdx.x & 63, rg = threadIdx.x >> 6, n = blockIdx.x * 256 + 4 * cq, r0 = blockIdx.y * 32;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (n < C) {
        for (int r = r0 + rg; r < min(r0 + 32, Rp); r += 4) {
            const f32x4 v = r < R ? *reinterpret_cast<const f32x4 *>(in + (size_t)r * C + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
            s += v;
            *reinterpret_cast<bf16x4 *>(out + (size_t)r * C + n) = (bf16x4){(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        }
    }
    if (part) {
        red4[rg][cq] = s;
        __syncthreads();
        if (rg == 0 && n < C && r0 < R) *reinterpret_cast<f32x4 *>(part + (size_t)blockIdx.y * C + n) = ((red4[0][cq] + red4[1][cq]) + red4[2][cq]) + red4[3][cq];
    }
}
// out (M, ldo) <- in (M, C), columns C..ldo-1 zero
__global__ static void k_pad_cols(const float *__restrict__ in, float *__restrict__ out, int M, int C, int ldo) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)M * ldo; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % ldo);
        out[i] = c < C ? in[(i / ldo) * C + c] : 0.f;
    }
}
// column sums of a (M, N) matrix, optionally of the elementwise product a .* b: part[chunk][n] over chunks of `rows` rows, then
// out[n] (+)= sum of chunks.  Fixed summation order (bit-reproducible steps): a workgroup = 64 columns x 4 row (chunk) groups, group g
// takes every fourth row (chunk) in order, the four partial sums are added in order through LDS.  (One thread per column walking all
// chunks serially -- the first form -- left a (9600, 256) bias gradient to ONE workgroup for 28 us, 474 times per training step.)
// Rows per chunk by the matrix height (colsum_chunk_rows): enough workgroups to fill the chip for the (9600, 256) bias gradients, not more
// than ~1000 partial rows for the frontend's (230 k, 256) ones.
static inline int colsum_chunk_rows(int M) { return M <= 16384 ? 32 : (M <= 65536 ? 64 : 256); }
__global__ __launch_bounds__(256) static void k_colsum_partial(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ part, int M, int N, int rows) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6, n = blockIdx.x * 64 + cl, chunk = blockIdx.y;
    const int r0 = chunk * rows, r1 = min(M, r0 + rows);
    float s = 0.f;
    if (n < N) {
        if (b) for (int r = r0 + rg; r < r1; r += 4) s = fmaf(a[(size_t)r * N + n], b[(size_t)r * N + n], s);
        else for (int r = r0 + rg; r < r1; r += 4) s += a[(size_t)r * N + n];
    }
    red[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && n < N) part[(size_t)chunk * N + n] = ((red[0][cl] + red[1][cl]) + red[2][cl]) + red[3][cl];
}
// the same for N % 4 == 0: a thread = four consecutive columns (16-byte loads: a wave reads 1 KB of a row, where the scalar form's 256-byte
// pieces reached 0.55 TB/s), a workgroup = a 256-column stripe x 4 row groups
__global__ __launch_bounds__(256) static void k_colsum_partial4(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ part, int M, int N, int rows) {
    __shared__ f32x4 red4[4][64];
    const int cq = threadIdx.x & 63, rg = threadIdx.x >> 6, n = blockIdx.x * 256 + 4 * cq, chunk = blockIdx.y;
    const int r0 = chunk * rows, r1 = min(M, r0 + rows);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
        if (b) for (int r = r0 + rg; r < r1; r += 4) s = __builtin_elementwise_fma(*reinterpret_cast<const f32x4 *>(a + (size_t)r * N + n), *reinterpret_cast<const f32x4 *>(b + (size_t)r * N + n), s);
        else for (int r = r0 + rg; r < r1; r += 4) s += *reinterpret_cast<const f32x4 *>(a + (size_t)r * N + n);
    }
    red4[rg][cq] = s;
    __syncthreads();
    if (rg == 0 && n < N) *reinterpret_cast<f32x4 *>(part + (size_t)chunk * N + n) = ((red4[0][cq] + red4[1][cq]) + red4[2][cq]) + red4[3][cq];
}
// Two matrices at once (a LayerNorm's d gamma and d beta operands; N % 4 == 0): part[chunk][2 N] = column sums of a1 | of a2 over the
// chunk's rows, the same arrangement as k_colsum_partial4; k_colsum_final_2 adds the chunks (16 chunk groups in order) into out1 / out2.
__global__ __launch_bounds__(256) static void k_colsum_partial4_2(const float *__restrict__ a1, const float *__restrict__ a2, float *__restrict__ part, int M, int N, int rows) {
    __shared__ f32x4 red4[4][64];
    const int cq = threadIdx.x & 63, rg = threadIdx.x >> 6, n2 = blockIdx.x * 256 + 4 * cq, chunk = blockIdx.y;      // n2: column of the (M, 2 N) pair
    const int r0 = chunk * rows, r1 = min(M, r0 + rows);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (n2 < 2 * N) {
        const float *a = n2 < N ? a1 + n2 : a2 + (n2 - N);
        for (int r = r0 + rg; r < r1; r += 4) s += *reinterpret_cast<const f32x4 *>(a + (size_t)r * N);
    }
    red4[rg][cq] = s;
    __syncthreads();
    if (rg == 0 && n2 < 2 * N) *reinterpret_cast<f32x4 *>(part + (size_t)chunk * 2 * N + n2) = ((red4[0][cq] + red4[1][cq]) + red4[2][cq]) + red4[3][cq];
}
// part[chunk][2 N] = column sums of a | of a .* a over the chunk's rows (BatchNorm's batch statistics in one pass; N % 4 == 0)
__global__ __launch_bounds__(256) static void k_colsum_partial4_sq(const float *__restrict__ a, float *__restrict__ part, int M, int N, int rows) {
    __shared__ f32x4 red4[4][64];
    const int cq = threadIdx.x & 63, rg = threadIdx.x >> 6, n2 = blockIdx.x * 256 + 4 * cq, chunk = blockIdx.y;
    const int r0 = chunk * rows, r1 = min(M, r0 + rows);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (n2 < N) { for (int r = r0 + rg; r < r1; r += 4) s += *reinterpret_cast<const f32x4 *>(a + (size_t)r * N + n2); }
    else if (n2 < 2 * N) { for (int r = r0 + rg; r < r1; r += 4) { const f32x4 v = *reinterpret_cast<const f32x4 *>(a + (size_t)r * N + n2 - N); s = __builtin_elementwise_fma(v, v, s); } }
    red4[rg][cq] = s;
    __syncthreads();
    if (rg == 0 && n2 < 2 * N) *reinterpret_cast<f32x4 *>(part + (size_t)chunk * 2 * N + n2) = ((red4[0][cq] + red4[1][cq]) + red4[2][cq]) + red4[3][cq];
}
// part[chunk][2 N] = column sums of a | of a .* b (BatchNorm backward: sum dy = d beta, sum dy xhat = d gamma; N % 4 == 0)
__global__ __launch_bounds__(256) static void k_colsum_partial4_ab(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ part, int M, int N, int rows) {
    __shared__ f32x4 red4[4][64];
    const int cq = threadIdx.x & 63, rg = threadIdx.x >> 6, n2 = blockIdx.x * 256 + 4 * cq, chunk = blockIdx.y;
    const int r0 = chunk * rows, r1 = min(M, r0 + rows);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (n2 < N) { for (int r = r0 + rg; r < r1; r += 4) s += *reinterpret_cast<const f32x4 *>(a + (size_t)r * N + n2); }
    else if (n2 < 2 * N) {
        for (int r = r0 + rg; r < r1; r += 4)
            s = __builtin_elementwise_fma(*reinterpret_cast<const f32x4 *>(a + (size_t)r * N + n2 - N), *reinterpret_cast<const f32x4 *>(b + (size_t)r * N + n2 - N), s);
    }
    red4[rg][cq] = s;
    __syncthreads();
    if (rg == 0 && n2 < 2 * N) *reinterpret_cast<f32x4 *>(part + (size_t)chunk * 2 * N + n2) = ((red4[0][cq] + red4[1][cq]) + red4[2][cq]) + red4[3][cq];
}
__global__ __launch_bounds__(256) static void k_colsum_final_2(const float *__restrict__ part, float *__restrict__ out1, float *__restrict__ out2, int chunks, int N) {
    __shared__ f32x4 red4[16][16];
    const int cq = threadIdx.x & 15, kg = threadIdx.x >> 4, n2 = blockIdx.x * 64 + 4 * cq;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (n2 < 2 * N) for (int k = kg; k < chunks; k += 16) s += *reinterpret_cast<const f32x4 *>(part + (size_t)k * 2 * N + n2);
    red4[kg][cq] = s;
    __syncthreads();
    if (kg == 0 && n2 < 2 * N) {
        f32x4 tot = red4[0][cq];
#pragma unroll
        for (int k = 1; k < 16; ++k) tot += red4[k][cq];
        *reinterpret_cast<f32x4 *>(n2 < N ? out1 + n2 : out2 + (n2 - N)) = tot;
    }
}
// Round 4: the finals of a step's gradient column sums (bias gradients, LayerNorm d gamma / d beta) as ONE launch at the end of the backward
// pass instead of one 4-workgroup launch each (330 of them, 7.6 us apiece: latency, not work).  A job = partial sums part[chunk * stride + n]
// of `chunks` chunks for N columns (N % 4 == 0) -> out[n]; workgroup b of the launch serves job j with first_block[j] <= b < first_block[j + 1],
// 64 columns each, 16 chunk groups added in a fixed order (bit-reproducible steps).
#define COCR_MAX_COLSUM_JOBS 2048
struct ColsumJob { const float *part; float *out; int stride, chunks, N, first_block; };
__global__ __launch_bounds__(256) static void k_colsum_final_jobs(const ColsumJob *__restrict__ jobs, int njobs) {
    __shared__ f32x4 red4[16][16];
    int lo = 0, hi = njobs - 1;                              // the last job whose first block is <= this block (uniform)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const ColsumJob j = jobs[lo];
    const int cq = threadIdx.x & 15, kg = threadIdx.x >> 4, n = ((int)blockIdx.x - j.first_block) * 64 + 4 * cq;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (n < j.N) for (int k = kg; k < j.chunks; k += 16) s += *reinterpret_cast<const f32x4 *>(j.part + (size_t)k * j.stride + n);
    red4[kg][cq] = s;
    __syncthreads();
    if (kg == 0 && n < j.N) {
        f32x4 tot = red4[0][cq];
#pragma unroll
        for (int k = 1; k < 16; ++k) tot += red4[k][cq];
        *reinterpret_cast<f32x4 *>(j.out + n) = tot;
    }
}
// out[n] (+)= sum over `chunks` of part[k][n]: the same 64 x 4 arrangement over the chunks
__global__ __launch_bounds__(256) static void k_colsum_final(const float *__restrict__ part, float *__restrict__ out, int chunks, int N, int accumulate) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, kg = threadIdx.x >> 6, n = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (n < N) for (int k = kg; k < chunks; k += 4) s += part[(size_t)k * N + n];
    red[kg][cl] = s;
    __syncthreads();
    if (kg == 0 && n < N) out[n] = (accumulate ? out[n] : 0.f) + (((red[0][cl] + red[1][cl]) + red[2][cl]) + red[3][cl]);
}
// the same for many chunks of a narrow matrix (N % 4 == 0): a thread = four columns, a workgroup = 64 columns x 16 chunk groups
__global__ __launch_bounds__(256) static void k_colsum_final4(const float *__restrict__ part, float *__restrict__ out, int chunks, int N, int accumulate) {
    __shared__ f32x4 red4[16][16];
    const int cq = threadIdx.x & 15, kg = threadIdx.x >> 4, n = blockIdx.x * 64 + 4 * cq;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (n < N) for (int k = kg; k < chunks; k += 16) s += *reinterpret_cast<const f32x4 *>(part + (size_t)k * N + n);
    red4[kg][cq] = s;
    __syncthreads();
    if (kg == 0 && n < N) {
        f32x4 t = red4[0][cq];
#pragma unroll
        for (int g2 = 1; g2 < 16; ++g2) t += red4[g2][cq];
#pragma unroll
        for (int q = 0; q < 4; ++q) out[n + q] = (accumulate ? out[n + q] : 0.f) + t[q];
    }
}

// ---- LayerNorm (eps 1e-5): one wave per row -----------------------------------------------------------------------------------------------
__global__ static void k_ln_fwd(const float *__restrict__ x, const float *__restrict__ g, const float *__restrict__ b, float *__restrict__ y,
                                float *__restrict__ mean, float *__restrict__ rstd, int M, int D) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float *xr = x + (size_t)row * D;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += xr[d];
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
    for (int d = lane; d < D; d += 64) { const float t = xr[d] - mu; q = fmaf(t, t, q); }
    const float rs = 1.0f / sqrtf(wave_sum(q) / (float)D + 1e-5f);
    for (int d = lane; d < D; d += 64) y[(size_t)row * D + d] = fmaf((xr[d] - mu) * rs, g[d], b[d]);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}
// dx (+)= rstd (dy g - mean_d(dy g) - xhat mean_d(dy g xhat));  dyxhat = dy .* xhat (for d gamma = column sum; d beta = column sum of dy)
__global__ static void k_ln_bwd(const float *__restrict__ dy, const float *__restrict__ x, const float *__restrict__ mean, const float *__restrict__ rstd,
                                const float *__restrict__ g, float *__restrict__ dx, float *__restrict__ dyxhat, int M, int D, int accumulate) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float mu = mean[row], rs = rstd[row];
    const float *xr = x + (size_t)row * D, *dr = dy + (size_t)row * D;
    float s1 = 0.f, s2 = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float xh = (xr[d] - mu) * rs, dg = dr[d] * g[d];
        s1 += dg;
        s2 = fmaf(dg, xh, s2);
    }
    const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
    for (int d = lane; d < D; d += 64) {
        const float xh = (xr[d] - mu) * rs, v = rs * (dr[d] * g[d] - c1 - xh * c2);
        const size_t o = (size_t)row * D + d;
        dyxhat[o] = dr[d] * xh;
        dx[o] = accumulate ? dx[o] + v : v;
    }
}

// ---- activations -----------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }
__global__ static void k_silu_fwd(const float *__restrict__ h, float *__restrict__ a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = h[i] * sigm(h[i]);
}
// a = drop(silu(h)); d <- drop(d) * silu'(h): the feed-forward module's hidden dropout folded into its activation, forward and backward
__global__ static void k_silu_fwd_drop(const float *__restrict__ h, float *__restrict__ a, size_t n, float p, unsigned long long seed, unsigned site) {
    const float sc = 1.0f / (1.0f - p);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = drop_apply(h[i] * sigm(h[i]), seed, site, i, p, sc);
}
__global__ static void k_silu_bwd_drop(const float *__restrict__ h, float *__restrict__ d, size_t n, float p, unsigned long long seed, unsigned site) {
    const float sc = 1.0f / (1.0f - p);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float s = sigm(h[i]);
        d[i] = drop_apply(d[i], seed, site, i, p, sc) * (s * fmaf(h[i], 1.0f - s, 1.0f));
    }
}
__global__ static void k_silu_bwd(const float *__restrict__ h, float *__restrict__ d, size_t n) {      // d <- d * silu'(h)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float s = sigm(h[i]);
        d[i] *= s * fmaf(h[i], 1.0f - s, 1.0f);
    }
}
// GLU over channels (convolution.py:139, dim=1 of (B, 2D, T)): g[m, c] = a[m, c] * sigmoid(a[m, D + c])
__global__ static void k_glu_fwd(const float *__restrict__ a, float *__restrict__ g, int M, int D) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)M * D; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / D, c = i - m * D;
        g[i] = a[m * 2 * D + c] * sigm(a[m * 2 * D + D + c]);
    }
}
__global__ static void k_glu_bwd(const float *__restrict__ a, const float *__restrict__ dg, float *__restrict__ da, int M, int D) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)M * D; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / D, c = i - m * D;
        const float v = a[m * 2 * D + c], s = sigm(a[m * 2 * D + D + c]);
        da[m * 2 * D + c] = dg[i] * s;
        da[m * 2 * D + D + c] = dg[i] * v * s * (1.0f - s);
    }
}
__global__ static void k_relu_bwd(const float *__restrict__ y, float *__restrict__ d, size_t n) {       // d <- y > 0 ? d : 0
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = y[i] > 0.f ? d[i] : 0.f;
}
__global__ static void k_relu(float *y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = fmaxf(y[i], 0.f);
}

// ---- relative-position attention (attention.py:72-113), train mode --------------------------------------------------------------------------------
// q, k, v (M, D) with head h in columns [h dh, (h+1) dh); P (2T-1, D): row r <-> relative position (T-1) - r, so score(i, j) uses row T-1-(i-j);
// u, vb (D) = u_bias / v_bias flattened (h, dh).  attn (N, h, T, T) is stored UNdropped; the dropout mask of the attention weights is
// regenerated from (seed, site, element index).
// forward: one wave per (line, head, query)
__global__ static void k_attn_fwd(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v, const float *__restrict__ P,
                                  const float *__restrict__ u, const float *__restrict__ vb, float *__restrict__ attn, float *__restrict__ ctx,
                                  long long total, int T, int H, int dh, float scale, float p, unsigned long long seed, unsigned site) {
    extern __shared__ float sm[];                      // per wave: qu[dh], qv[dh]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const long long row = (long long)blockIdx.x * wpb + wave;       // (b * H + h) * T + i
    if (row >= total) return;
    const int D = H * dh;
    const int i = (int)(row % T), bh = (int)(row / T), h = bh % H, b = bh / H;
    float *qu = sm + wave * 2 * dh, *qv = qu + dh;
    const float *qr = q + ((size_t)b * T + i) * D + h * dh;
    for (int d = lane; d < dh; d += 64) { qu[d] = qr[d] + u[h * dh + d]; qv[d] = qr[d] + vb[h * dh + d]; }
    __builtin_amdgcn_wave_barrier();
    float *ar = attn + (size_t)row * T;
    float mx = -INFINITY;
    for (int j = lane; j < T; j += 64) {
        const float *kr = k + ((size_t)b * T + j) * D + h * dh, *pr = P + (size_t)(T - 1 - (i - j)) * D + h * dh;
        float s = 0.f;
        if ((dh & 3) == 0 && (D & 3) == 0) {           // 16-byte loads of the two rows (a lane walks its own key's row: every load instruction touches 64 lines)
            const float4 *k4 = reinterpret_cast<const float4 *>(kr), *p4 = reinterpret_cast<const float4 *>(pr);
#pragma unroll 4
            for (int d4 = 0; d4 < (dh >> 2); ++d4) {
                const float4 kk = k4[d4], pp = p4[d4];
                const int d = 4 * d4;
                s = fmaf(qu[d], kk.x, fmaf(qv[d], pp.x, s));
                s = fmaf(qu[d + 1], kk.y, fmaf(qv[d + 1], pp.y, s));
                s = fmaf(qu[d + 2], kk.z, fmaf(qv[d + 2], pp.z, s));
                s = fmaf(qu[d + 3], kk.w, fmaf(qv[d + 3], pp.w, s));
            }
        } else {
            for (int d = 0; d < dh; ++d) s = fmaf(qu[d], kr[d], fmaf(qv[d], pr[d], s));
        }
        s *= scale;
        ar[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < T; j += 64) { const float e = expf(ar[j] - mx); ar[j] = e; sum += e; }
    const float inv = 1.0f / wave_sum(sum);
    for (int j = lane; j < T; j += 64) ar[j] *= inv;
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    const float sc = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    for (int d = lane; d < dh; d += 64) {
        float acc = 0.f;
#pragma unroll 8
        for (int j = 0; j < T; ++j) {
            float a = ar[j];
            if (p > 0.f) a = drop_keep(seed, site, (unsigned long long)row * T + j, p) ? a * sc : 0.f;
            acc = fmaf(a, v[((size_t)b * T + j) * D + h * dh + d], acc);
        }
        ctx[((size_t)b * T + i) * D + h * dh + d] = acc;
    }
}
// backward, rows: one wave per (line, head, query): ds (overwrites `dsb` row), du_part[i] = sum_j ds k_j, dvb_part[i] = sum_j ds P_rel  (dq = their sum)
__global__ static void k_attn_bwd_rows(const float *__restrict__ dctx, const float *__restrict__ k, const float *__restrict__ v, const float *__restrict__ P,
                                       const float *__restrict__ attn, float *__restrict__ dsb, float *__restrict__ du_part, float *__restrict__ dvb_part,
                                       long long total, int T, int H, int dh, float scale, float p, unsigned long long seed, unsigned site) {
    extern __shared__ float sm[];                      // per wave: dctx row [dh]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const long long row = (long long)blockIdx.x * wpb + wave;
    if (row >= total) return;
    const int D = H * dh, i = (int)(row % T), bh = (int)(row / T), h = bh % H, b = bh / H;
    float *dc = sm + wave * dh;
    for (int d = lane; d < dh; d += 64) dc[d] = dctx[((size_t)b * T + i) * D + h * dh + d];
    __builtin_amdgcn_wave_barrier();
    const float *ar = attn + (size_t)row * T;
    float *dr = dsb + (size_t)row * T;
    const float sc = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    float delta = 0.f;
    for (int j = lane; j < T; j += 64) {
        const float *vr = v + ((size_t)b * T + j) * D + h * dh;
        float da = 0.f;
        if ((dh & 3) == 0 && (D & 3) == 0) {
            const float4 *v4 = reinterpret_cast<const float4 *>(vr);
#pragma unroll 4
            for (int d4 = 0; d4 < (dh >> 2); ++d4) {
                const float4 vv = v4[d4];
                const int d = 4 * d4;
                da = fmaf(dc[d], vv.x, da); da = fmaf(dc[d + 1], vv.y, da); da = fmaf(dc[d + 2], vv.z, da); da = fmaf(dc[d + 3], vv.w, da);
            }
        } else {
            for (int d = 0; d < dh; ++d) da = fmaf(dc[d], vr[d], da);
        }
        if (p > 0.f) da = drop_keep(seed, site, (unsigned long long)row * T + j, p) ? da * sc : 0.f;      // d loss / d (undropped attention weight)
        dr[j] = da;
        delta = fmaf(ar[j], da, delta);
    }
    delta = wave_sum(delta);
    for (int j = lane; j < T; j += 64) dr[j] = ar[j] * (dr[j] - delta) * scale;
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    for (int d = lane; d < dh; d += 64) {
        float a1 = 0.f, a2 = 0.f;
#pragma unroll 8
        for (int j = 0; j < T; ++j) {
            const float ds = dr[j];
            a1 = fmaf(ds, k[((size_t)b * T + j) * D + h * dh + d], a1);
            a2 = fmaf(ds, P[(size_t)(T - 1 - (i - j)) * D + h * dh + d], a2);
        }
        du_part[((size_t)b * T + i) * D + h * dh + d] = a1;
        dvb_part[((size_t)b * T + i) * D + h * dh + d] = a2;
    }
}
// backward, columns: one wave per (line, head, key j): dk_j = sum_i ds_ij (q_i + u), dv_j = sum_i attn_dropped_ij dctx_i
__global__ static void k_attn_bwd_cols(const float *__restrict__ dctx, const float *__restrict__ q, const float *__restrict__ u, const float *__restrict__ attn,
                                       const float *__restrict__ dsb, float *__restrict__ dk, float *__restrict__ dv,
                                       long long total, int T, int H, int dh, float p, unsigned long long seed, unsigned site) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const long long col = (long long)blockIdx.x * wpb + wave;       // (b * H + h) * T + j
    if (col >= total) return;
    const int D = H * dh, j = (int)(col % T), bh = (int)(col / T), h = bh % H, b = bh / H;
    const float sc = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    for (int d = lane; d < dh; d += 64) {
        float a1 = 0.f, a2 = 0.f;
        const float ud = u[h * dh + d];
#pragma unroll 8
        for (int i = 0; i < T; ++i) {
            const size_t e = ((size_t)bh * T + i) * T + j;
            a1 = fmaf(dsb[e], q[((size_t)b * T + i) * D + h * dh + d] + ud, a1);
            float a = attn[e];
            if (p > 0.f) a = drop_keep(seed, site, e, p) ? a * sc : 0.f;
            a2 = fmaf(a, dctx[((size_t)b * T + i) * D + h * dh + d], a2);
        }
        dk[((size_t)b * T + j) * D + h * dh + d] = a1;
        dv[((size_t)b * T + j) * D + h * dh + d] = a2;
    }
}
// backward, positional table: one wave per (row r, head) of ONE line (blockIdx.y): part[b][r, h, :] = sum over the diagonal i - j = T-1-r of
// ds_ij (q_i + vb); summed over the lines by k_colsum_final
__global__ static void k_attn_bwd_pos(const float *__restrict__ q, const float *__restrict__ vb, const float *__restrict__ dsb, float *__restrict__ part,
                                      int N, int T, int H, int dh) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const int id = blockIdx.x * wpb + wave, b = blockIdx.y;             // id = r * H + h
    if (id >= (2 * T - 1) * H) return;
    const int D = H * dh, r = id / H, h = id - r * H, off = T - 1 - r;      // i - j = off
    const int i0 = max(0, off), i1 = min(T, T + off);
    for (int d = lane; d < dh; d += 64) {
        float acc = 0.f;
        const float vd = vb[h * dh + d];
#pragma unroll 8
        for (int i = i0; i < i1; ++i)
            acc = fmaf(dsb[(((size_t)b * H + h) * T + i) * T + (i - off)], q[((size_t)b * T + i) * D + h * dh + d] + vd, acc);
        part[((size_t)b * (2 * T - 1) + r) * D + h * dh + d] = acc;
    }
}

// ---- the same attention as batched matrix products (d_head a multiple of 32) ---------------------------------------------------------------
// The kernels above give one wave a query (or key) row and walk the other dimension with scalar loads: 1.1 ms forward + 2.2 ms backward
// per block at 32 x 300 frames, 45 % of the training step.  In this form every product of the module is a batched exact-fp32 MFMA GEMM
// over the (line, head) pairs (gemm.hip.h: launch_gemm_batched_f32), with elementwise / transposing kernels between them:
//   forward : S = (q + u) K^T ; Rm = (q + vb) P_h^T (all 2T-1 relative positions) ; attn = softmax((S + shift(Rm)) scale) ; ctx = drop(attn) V
//   backward: dA = dctx V^T ; ds = attn (drop'(dA) - rowsum(attn drop'(dA))) scale ; dV = drop(attn)^T dctx ; d(q+u) = ds K ;
//             dK = ds^T (q + u) ; dR = shift^-1(ds) ; d(q+vb) = dR P_h ; dP_h = sum_lines dR^T (q + vb)
// Square matrices are stored with rows of Tk = round_up(T, 32) floats, the relative-position ones with Rk = round_up(2T-1, 32) (the
// products' K dimension; padding is zero).  The dropout mask is the same function of (seed, site, element) as above.
__global__ static void k_attn_qu_qv(const float *__restrict__ q, const float *__restrict__ u, const float *__restrict__ vb, float *__restrict__ qu,
                                    float *__restrict__ qv, size_t n, int D) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % D);
        const float x = q[i];
        qu[i] = x + u[c]; qv[i] = x + vb[c];
    }
}
// Batched transpose with zero padding: out[z][c][r] = in[z][r][c] for r < R, c < C; out[z][c][r] = 0 for R <= r < Rpad.  Input batch z =
// (zb, zh) = (z / zdiv, z % zdiv) at element offset zb * izb + zh * izh, row stride ldi; output batch at z * oz, row stride ldo.
// drop_T > 0: the input is the attention matrix of (line, head) z (drop_T = T): the dropout of the attention weights is applied on the way.
__global__ static void k_btranspose(const float *__restrict__ in, float *__restrict__ out, int R, int C, long long ldi, long long ldo, int Rpad, int zdiv,
                                    long long izb, long long izh, long long oz, int drop_T, float p, unsigned long long seed, unsigned site) {
    __shared__ float tile[32][33];
    const int z = blockIdx.z, zb = z / zdiv, zh = z - zb * zdiv;
    const float *src = in + zb * izb + zh * izh;
    float *dst = out + (long long)z * oz;
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float sc = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    for (int j = ty; j < 32; j += 8) {
        float v = 0.f;
        if (r0 + j < R && c0 + tx < C) {
            v = src[(long long)(r0 + j) * ldi + c0 + tx];
            if (drop_T > 0 && p > 0.f) v = drop_keep(seed, site, ((unsigned long long)z * drop_T + (r0 + j)) * drop_T + (c0 + tx), p) ? v * sc : 0.f;
        }
        tile[j][tx] = v;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < C && r0 + tx < Rpad) dst[(long long)(c0 + j) * ldo + r0 + tx] = tile[tx][j];
}
// attn row (z, i) <- softmax_j((S[j] + Rm[T-1-i+j]) scale), in place over S; columns T..Tk-1 zero; ad (nullable) <- the dropped weights
__global__ static void k_attn_softmax(float *__restrict__ S, const float *__restrict__ Rm, float *__restrict__ ad, long long rows, int T, int Tk, int Rk,
                                      float scale, float p, unsigned long long seed, unsigned site) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const long long row = (long long)blockIdx.x * wpb + wave;       // z * T + i
    if (row >= rows) return;
    const int i = (int)(row % T);
    float *sr = S + row * Tk;
    const float *rr = Rm + row * Rk + (T - 1 - i);
    float mx = -INFINITY;
    for (int j = lane; j < T; j += 64) { const float s = (sr[j] + rr[j]) * scale; sr[j] = s; mx = fmaxf(mx, s); }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < T; j += 64) { const float e = expf(sr[j] - mx); sr[j] = e; sum += e; }
    const float inv = 1.0f / wave_sum(sum), sc = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    for (int j = lane; j < Tk; j += 64) {
        const float a = j < T ? sr[j] * inv : 0.f;
        sr[j] = a;
        if (ad) ad[row * Tk + j] = (j < T && p > 0.f) ? (drop_keep(seed, site, (unsigned long long)row * T + j, p) ? a * sc : 0.f) : a;
    }
}
// ds row (z, i) <- attn (da - sum_j attn da) scale with da = d loss / d (undropped weight) = drop'(dA); columns T..Tk-1 zero.  In place over dA.
__global__ static void k_attn_softmax_bwd(float *__restrict__ dA, const float *__restrict__ attn, long long rows, int T, int Tk, float scale, float p,
                                          unsigned long long seed, unsigned site) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const long long row = (long long)blockIdx.x * wpb + wave;
    if (row >= rows) return;
    float *dr = dA + row * Tk;
    const float *ar = attn + row * Tk;
    const float sc = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    float delta = 0.f;
    for (int j = lane; j < T; j += 64) {
        float da = dr[j];
        if (p > 0.f) da = drop_keep(seed, site, (unsigned long long)row * T + j, p) ? da * sc : 0.f;
        dr[j] = da;
        delta = fmaf(ar[j], da, delta);
    }
    delta = wave_sum(delta);
    for (int j = lane; j < Tk; j += 64) dr[j] = j < T ? ar[j] * (dr[j] - delta) * scale : 0.f;
}
// dR[z][i][r] = ds[z][i][r - (T-1-i)] where that key index is in [0, T), else 0 (r < Rk): the inverse of the forward's shift
__global__ static void k_attn_unshift(const float *__restrict__ ds, float *__restrict__ dR, long long rows, int T, int Tk, int Rk) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const long long row = (long long)blockIdx.x * wpb + wave;
    if (row >= rows) return;
    const int i = (int)(row % T), off = T - 1 - i;
    const float *sr = ds + row * Tk;
    float *dr = dR + row * Rk;
    for (int r = lane; r < Rk; r += 64) { const int j = r - off; dr[r] = (j >= 0 && j < T) ? sr[j] : 0.f; }
}

// ---- conv module: depthwise conv along time (convolution.py:140; cross-correlation, zero padding at the ends of the PADDED batch rows) --------
// Kernel sizes up to 32 (the reference's 31): a thread = one channel of one line over a chunk of frames, its taps in registers, accumulation
// in tap order.  grid = (ceil(D / 256), frame chunks, lines).  (The first forms -- a flat grid-stride loop with two 64-bit divisions per element
// and a tap load per multiply, strided by K across the lanes -- took 125 us per call on (9600, 256): 12 % of the training step.)
#define COCR_DW_TC 16             // frames per workgroup (forward / input gradient, and per partial sum of the tap gradient)
#define COCR_DW_WC COCR_DW_TC
// KC: the kernel size at compile time (31: the reference's), 0: K <= 32 at run time.  Frames whose whole tap range lies inside the line take
// a branch-free path (the loads of a frame are then requested together; a uniform range test per tap kept them one behind the other).
template <bool FLIP, int KC>      // FLIP false: out[t] = sum_tau w[tau] in[t + tau - pad];  true: out[t] = sum_tau w[tau] in[t - tau + pad] (input gradient)
__global__ __launch_bounds__(256) static void k_dw1d_rows(const float *__restrict__ in, const float *__restrict__ w, float *__restrict__ out, int N, int T, int D, int Krt) {
    const int K = KC ? KC : Krt;
    constexpr int KU = KC ? KC : 32;
    const int c = blockIdx.x * 256 + threadIdx.x, b = blockIdx.z, t0 = blockIdx.y * COCR_DW_TC, t1 = min(T, t0 + COCR_DW_TC), pad = (K - 1) / 2;
    if (c >= D) return;
    float wt[KU];
#pragma unroll
    for (int tau = 0; tau < KU; ++tau) wt[tau] = tau < K ? w[c * K + tau] : 0.f;
    const float *base = in + (size_t)b * T * D + c;
    for (int t = t0; t < t1; ++t) {
        float acc = 0.f;
        if (KC && t >= pad && t + pad < T) {
            const float *p0 = base + (size_t)(FLIP ? t + pad : t - pad) * D;
            float x[KU];
#pragma unroll
            for (int tau = 0; tau < KU; ++tau) x[tau] = FLIP ? p0[-(ptrdiff_t)tau * D] : p0[(size_t)tau * D];
#pragma unroll
            for (int tau = 0; tau < KU; ++tau) acc = fmaf(wt[tau], x[tau], acc);
        } else {
#pragma unroll
            for (int tau = 0; tau < KU; ++tau)
                if (tau < K) {
                    const int tt = FLIP ? t - tau + pad : t + tau - pad;
                    if (tt >= 0 && tt < T) acc = fmaf(wt[tau], base[(size_t)tt * D], acc);
                }
        }
        out[((size_t)b * T + t) * D + c] = acc;
    }
}
// dw[c, tau] = sum over lines and frames of dout[b, t, c] g[b, t + tau - pad, c]: partial sums per (line, chunk of COCR_DW_WC frames),
// part[(b nch + chunk)][c K + tau], summed in order by k_colsum_final
template <int KC>
__global__ __launch_bounds__(256) static void k_dw1d_bwd_w(const float *__restrict__ dout, const float *__restrict__ g, float *__restrict__ part, int N, int T, int D, int Krt) {
    const int K = KC ? KC : Krt;
    constexpr int KU = KC ? KC : 32;
    const int c = blockIdx.x * 256 + threadIdx.x, b = blockIdx.z, t0 = blockIdx.y * COCR_DW_WC, t1 = min(T, t0 + COCR_DW_WC), pad = (K - 1) / 2;
    if (c >= D) return;
    float acc[KU];
#pragma unroll
    for (int tau = 0; tau < KU; ++tau) acc[tau] = 0.f;
    const float *gb = g + (size_t)b * T * D + c;
    for (int t = t0; t < t1; ++t) {
        const float d = dout[((size_t)b * T + t) * D + c];
        if (KC && t >= pad && t + pad < T) {
            const float *p0 = gb + (size_t)(t - pad) * D;
            float x[KU];
#pragma unroll
            for (int tau = 0; tau < KU; ++tau) x[tau] = p0[(size_t)tau * D];
#pragma unroll
            for (int tau = 0; tau < KU; ++tau) acc[tau] = fmaf(d, x[tau], acc[tau]);
        } else {
#pragma unroll
            for (int tau = 0; tau < KU; ++tau)
                if (tau < K) {
                    const int tt = t + tau - pad;
                    if (tt >= 0 && tt < T) acc[tau] = fmaf(d, gb[(size_t)tt * D], acc[tau]);
                }
        }
    }
    float *pr = part + ((size_t)b * gridDim.y + blockIdx.y) * D * K + (size_t)c * K;
#pragma unroll
    for (int tau = 0; tau < KU; ++tau) if (tau < K) pr[tau] = acc[tau];
}
// (any kernel size: the flat forms)
__global__ static void k_dw1d_fwd_flat(const float *__restrict__ g, const float *__restrict__ w, float *__restrict__ out, int N, int T, int D, int K, int flip) {
    const int pad = (K - 1) / 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)N * T * D; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % D), t = (int)((i / D) % T);
        const size_t base = i - (size_t)t * D;
        float acc = 0.f;
        for (int tau = 0; tau < K; ++tau) {
            const int tt = flip ? t - tau + pad : t + tau - pad;
            if (tt >= 0 && tt < T) acc = fmaf(w[c * K + tau], g[base + (size_t)tt * D], acc);
        }
        out[i] = acc;
    }
}
// thread = (c, tau) of ONE line (blockIdx.z), frames in order; part[b][c K + tau]
__global__ static void k_dw1d_bwd_w_flat(const float *__restrict__ dout, const float *__restrict__ g, float *__restrict__ part, int N, int T, int D, int K) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, tau = blockIdx.y, b = blockIdx.z, pad = (K - 1) / 2;
    if (c >= D) return;
    float acc = 0.f;
    for (int t = max(0, pad - tau); t < min(T, T + pad - tau); ++t)
        acc = fmaf(dout[((size_t)b * T + t) * D + c], g[((size_t)b * T + t + tau - pad) * D + c], acc);
    part[(size_t)b * D * K + c * K + tau] = acc;
}

// ---- BatchNorm1d, train mode (convolution.py:141) -------------------------------------------------------------------------------------------------
// from column sums of x and x^2 over the M positions: batch mean, 1/sqrt(biased var + eps); running statistics (momentum 0.1, unbiased variance)
__global__ static void k_bn_finalize(const float *__restrict__ sum, const float *__restrict__ sumsq, int M, int D, float *__restrict__ mean, float *__restrict__ rstd,
                                     float *__restrict__ running_mean, float *__restrict__ running_var, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= D) return;
    const float mu = sum[c] / (float)M, var = fmaxf(sumsq[c] / (float)M - mu * mu, 0.f);
    mean[c] = mu;
    rstd[c] = 1.0f / sqrtf(var + 1e-5f);
    running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mu;
    running_var[c] = (1.0f - momentum) * running_var[c] + momentum * var * ((float)M / (float)max(M - 1, 1));
}
__global__ static void k_bn_apply(const float *__restrict__ x, const float *__restrict__ mean, const float *__restrict__ rstd, const float *__restrict__ g,
                                  const float *__restrict__ b, float *__restrict__ xhat, float *__restrict__ y, int M, int D) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)M * D; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % D);
        const float xh = (x[i] - mean[c]) * rstd[c];
        xhat[i] = xh;
        y[i] = fmaf(xh, g[c], b[c]);
    }
}
// the same with the SiLU behind it (convolution.py:141-142): xhat, y = BatchNorm output, a = silu(y) in one pass
__global__ static void k_bn_apply_silu(const float *__restrict__ x, const float *__restrict__ mean, const float *__restrict__ rstd, const float *__restrict__ g,
                                       const float *__restrict__ b, float *__restrict__ xhat, float *__restrict__ y, float *__restrict__ a, int M, int D) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)M * D; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % D);
        const float xh = (x[i] - mean[c]) * rstd[c];
        xhat[i] = xh;
        const float v = fmaf(xh, g[c], b[c]);
        y[i] = v;
        a[i] = v * sigm(v);
    }
}
// dx = g rstd / M (M dy - sum dy - xhat sum(dy xhat))
__global__ static void k_bn_bwd(const float *__restrict__ dy, const float *__restrict__ xhat, const float *__restrict__ g, const float *__restrict__ rstd,
                                const float *__restrict__ sum_dy, const float *__restrict__ sum_dyxh, float *__restrict__ dx, int M, int D) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)M * D; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % D);
        dx[i] = g[c] * rstd[c] * (dy[i] - (sum_dy[c] + xhat[i] * sum_dyxh[c]) / (float)M);
    }
}

// ---- frontend convolutions (convolution.py:192-213), channel-last activations (n, t, f, c) --------------------------------------------------------
// Work decomposition of all of them: a 256-thread block = one or a few rows (n, t); thread = four consecutive channels c4 = tid % (C / 4)
// (16-byte accesses, a wave covers 1 KB of one position) x a position phase ph = tid / (C / 4) that strides over f.  A thread's taps stay in
// registers over its positions; no 64-bit index arithmetic per element (the first forms of these kernels -- a grid-stride loop over flat
// 64-bit indices with three divisions per element -- ran at 0.4 - 0.5 TB/s: 2 ms each on the 944 MB of conv.0's output at 32 x 96 x 1200).
// C % 4 == 0 and C <= 1024 (cocr_train_begin checks).
#define COCR_CV_ROWS 8            // rows (n, t) per block of the weight-gradient kernels = per partial sum
// conv.0: Z1[n, t1, f1, c] = relu(b0[c] + sum_{dt,df} w0[c, dt, df] X[n, 2 f1 + df - 1, 2 t1 + dt - 1])    (X (N, H, W); zero outside)
__global__ __launch_bounds__(256) static void k_conv0_fwd(const float *__restrict__ X, const float *__restrict__ w0, const float *__restrict__ b0,
                                                          float *__restrict__ Z1, int N, int H, int W, int T1, int F1, int C) {
    const unsigned C4 = (unsigned)C >> 2, FP = 256u / C4, c4 = threadIdx.x % C4, ph = threadIdx.x / C4;
    if (ph >= FP) return;
    const unsigned row = blockIdx.x, n = row / (unsigned)T1, t = row - n * (unsigned)T1, c = 4 * c4;
    float w[4][9];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < 9; ++k) w[q][k] = w0[(c + q) * 9 + k];
    const f32x4 bias = *reinterpret_cast<const f32x4 *>(b0 + c);
    const float *xb = X + (size_t)n * H * W;
    float *zrow = Z1 + (size_t)row * F1 * C + c;
    for (unsigned f = ph; f < (unsigned)F1; f += FP) {
        f32x4 acc = bias;
#pragma unroll
        for (int dt = 0; dt < 3; ++dt)
#pragma unroll
            for (int df = 0; df < 3; ++df) {
                const int col = 2 * (int)t + dt - 1, r = 2 * (int)f + df - 1;
                if (col >= 0 && col < W && r >= 0 && r < H) {
                    const float x = xb[(size_t)r * W + col];
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] = fmaf(w[q][dt * 3 + df], x, acc[q]);
                }
            }
        *reinterpret_cast<f32x4 *>(zrow + (size_t)f * C) = (f32x4){fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f), fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f)};
    }
}
// block-level tail of the two weight-gradient kernels: acc[k][q] of every thread -> partial sums over the block's position phases,
// part[(chunk * 10 + k) * C + c] (k < 9: taps, k == 9: bias); phases are added in ascending order (deterministic)
__device__ __forceinline__ void conv_w_block_sum(float (&acc)[10][4], float *__restrict__ part, unsigned chunk, unsigned C, unsigned C4, unsigned FP,
                                                 unsigned c4, unsigned ph) {
    extern __shared__ __attribute__((aligned(16))) float cw_red[];          // [FP][10][C]
    if (ph < FP) {
#pragma unroll
        for (int k = 0; k < 10; ++k) *reinterpret_cast<f32x4 *>(cw_red + ((size_t)ph * 10 + k) * C + 4 * c4) = (f32x4){acc[k][0], acc[k][1], acc[k][2], acc[k][3]};
    }
    __syncthreads();
    for (unsigned i = threadIdx.x; i < 10 * C; i += 256) {
        float sum = cw_red[i];
        for (unsigned p2 = 1; p2 < FP; ++p2) sum += cw_red[(size_t)p2 * 10 * C + i];
        part[(size_t)chunk * 10 * C + i] = sum;
    }
}
// d w0[c, tap] and d b0[c] partial sums over COCR_CV_ROWS rows (dZ1 already masked by the ReLU)
__global__ __launch_bounds__(256) static void k_conv0_bwd_w(const float *__restrict__ dZ1, const float *__restrict__ X, float *__restrict__ part,
                                                            int N, int H, int W, int T1, int F1, int C) {
    const unsigned C4 = (unsigned)C >> 2, FP = 256u / C4, c4 = threadIdx.x % C4, ph = threadIdx.x / C4, c = 4 * c4;
    float acc[10][4];
#pragma unroll
    for (int k = 0; k < 10; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[k][q] = 0.f;
    const unsigned row0 = blockIdx.x * COCR_CV_ROWS, row1 = min((unsigned)N * (unsigned)T1, row0 + COCR_CV_ROWS);
    if (ph < FP) {
        for (unsigned row = row0; row < row1; ++row) {
            const unsigned n = row / (unsigned)T1, t = row - n * (unsigned)T1;
            const float *xb = X + (size_t)n * H * W;
            const float *drow = dZ1 + (size_t)row * F1 * C + c;
            for (unsigned f = ph; f < (unsigned)F1; f += FP) {
                const f32x4 d = *reinterpret_cast<const f32x4 *>(drow + (size_t)f * C);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[9][q] += d[q];
#pragma unroll
                for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                    for (int df = 0; df < 3; ++df) {
                        const int col = 2 * (int)t + dt - 1, r = 2 * (int)f + df - 1;
                        if (col >= 0 && col < W && r >= 0 && r < H) {
                            const float x = xb[(size_t)r * W + col];
#pragma unroll
                            for (int q = 0; q < 4; ++q) acc[dt * 3 + df][q] = fmaf(d[q], x, acc[dt * 3 + df][q]);
                        }
                    }
            }
        }
    }
    conv_w_block_sum(acc, part, blockIdx.x, (unsigned)C, C4, FP, c4, ph);
}
// depthwise 3x3 stride 2: Zo[n, t, f, c] = b[c] + sum w[c, dt, df] Zi[n, 2 t + dt - 1, 2 f + df - 1, c]
__global__ __launch_bounds__(256) static void k_dw3_fwd(const float *__restrict__ Zi, const float *__restrict__ w, const float *__restrict__ b,
                                                        float *__restrict__ Zo, int N, int Ti, int Fi, int To, int Fo, int C) {
    const unsigned C4 = (unsigned)C >> 2, FP = 256u / C4, c4 = threadIdx.x % C4, ph = threadIdx.x / C4;
    if (ph >= FP) return;
    const unsigned row = blockIdx.x, n = row / (unsigned)To, t = row - n * (unsigned)To, c = 4 * c4;
    float wt[4][9];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < 9; ++k) wt[q][k] = w[(c + q) * 9 + k];
    const f32x4 bias = *reinterpret_cast<const f32x4 *>(b + c);
    for (unsigned f = ph; f < (unsigned)Fo; f += FP) {
        f32x4 acc = bias;
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
            const int ti = 2 * (int)t + dt - 1;
            if (ti < 0 || ti >= Ti) continue;
#pragma unroll
            for (int df = 0; df < 3; ++df) {
                const int fi = 2 * (int)f + df - 1;
                if (fi < 0 || fi >= Fi) continue;
                const f32x4 z = *reinterpret_cast<const f32x4 *>(Zi + (((size_t)n * Ti + ti) * Fi + fi) * C + c);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = fmaf(wt[q][dt * 3 + df], z[q], acc[q]);
            }
        }
        *reinterpret_cast<f32x4 *>(Zo + ((size_t)row * Fo + f) * C + c) = acc;
    }
}
// d Zi[n, ti, fi, c] = sum over the outputs (t, f) with 2 t + dt - 1 = ti, 2 f + df - 1 = fi of w[c, dt, df] dZo[n, t, f, c];
// `relu_of`: Zi itself -- the result is masked by Zi > 0 (the ReLU that produced Zi: saves the separate pass over the largest tensor)
__global__ __launch_bounds__(256) static void k_dw3_bwd_in(const float *__restrict__ dZo, const float *__restrict__ w, float *__restrict__ dZi,
                                                           const float *__restrict__ relu_of, int N, int Ti, int Fi, int To, int Fo, int C) {
    const unsigned C4 = (unsigned)C >> 2, FP = 256u / C4, c4 = threadIdx.x % C4, ph = threadIdx.x / C4;
    if (ph >= FP) return;
    const unsigned row = blockIdx.x, n = row / (unsigned)Ti, ti = row - n * (unsigned)Ti, c = 4 * c4;
    float wt[4][9];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < 9; ++k) wt[q][k] = w[(c + q) * 9 + k];
    for (unsigned fi = ph; fi < (unsigned)Fi; fi += FP) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
            const int t2 = (int)ti + 1 - dt;             // 2 t + dt - 1 = ti
            if (t2 < 0 || (t2 & 1) || (t2 >> 1) >= To) continue;
#pragma unroll
            for (int df = 0; df < 3; ++df) {
                const int f2 = (int)fi + 1 - df;
                if (f2 < 0 || (f2 & 1) || (f2 >> 1) >= Fo) continue;
                const f32x4 d = *reinterpret_cast<const f32x4 *>(dZo + (((size_t)n * To + (t2 >> 1)) * Fo + (f2 >> 1)) * C + c);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = fmaf(wt[q][dt * 3 + df], d[q], acc[q]);
            }
        }
        const size_t o = ((size_t)row * Fi + fi) * C + c;
        if (relu_of) {
            const f32x4 z = *reinterpret_cast<const f32x4 *>(relu_of + o);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = z[q] > 0.f ? acc[q] : 0.f;
        }
        *reinterpret_cast<f32x4 *>(dZi + o) = acc;
    }
}
__global__ __launch_bounds__(256) static void k_dw3_bwd_w(const float *__restrict__ dZo, const float *__restrict__ Zi, float *__restrict__ part,
                                                          int N, int Ti, int Fi, int To, int Fo, int C) {
    const unsigned C4 = (unsigned)C >> 2, FP = 256u / C4, c4 = threadIdx.x % C4, ph = threadIdx.x / C4, c = 4 * c4;
    float acc[10][4];
#pragma unroll
    for (int k = 0; k < 10; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[k][q] = 0.f;
    const unsigned row0 = blockIdx.x * COCR_CV_ROWS, row1 = min((unsigned)N * (unsigned)To, row0 + COCR_CV_ROWS);
    if (ph < FP) {
        for (unsigned row = row0; row < row1; ++row) {
            const unsigned n = row / (unsigned)To, t = row - n * (unsigned)To;
            for (unsigned f = ph; f < (unsigned)Fo; f += FP) {
                const f32x4 d = *reinterpret_cast<const f32x4 *>(dZo + ((size_t)row * Fo + f) * C + c);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[9][q] += d[q];
#pragma unroll
                for (int dt = 0; dt < 3; ++dt) {
                    const int ti = 2 * (int)t + dt - 1;
                    if (ti < 0 || ti >= Ti) continue;
#pragma unroll
                    for (int df = 0; df < 3; ++df) {
                        const int fi = 2 * (int)f + df - 1;
                        if (fi < 0 || fi >= Fi) continue;
                        const f32x4 z = *reinterpret_cast<const f32x4 *>(Zi + (((size_t)n * Ti + ti) * Fi + fi) * C + c);
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[dt * 3 + df][q] = fmaf(d[q], z[q], acc[dt * 3 + df][q]);
                    }
                }
            }
        }
    }
    conv_w_block_sum(acc, part, blockIdx.x, (unsigned)C, C4, FP, c4, ph);
}
// partials [chunk][10][C] -> dw[c][9] and db[c]: a block per 64 (k, c) entries, its 4 waves take every fourth chunk; fixed order
__global__ __launch_bounds__(256) static void k_conv_w_final(const float *__restrict__ part, int chunks, int C, float *__restrict__ dw, float *__restrict__ db) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, i = blockIdx.x * 64 + lane;      // i = k * C + c
    float s = 0.f;
    if (i < 10 * C)
        for (int ch = wv; ch < chunks; ch += 4) s += part[(size_t)ch * 10 * C + i];
    red[wv][lane] = s;
    __syncthreads();
    if (wv == 0 && i < 10 * C) {
        const float tot = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        const int k = i / C, c = i - k * C;
        if (k < 9) dw[c * 9 + k] = tot; else db[c] = tot;
    }
}
// (n, t, c, f) <-> (n, t, f, c): the reference flattens the frontend output channel-major (convolution.py:235-236: feature index c F + f).
// A block per row (n, t): the F x C tile goes through LDS, both sides coalesced.  F * C * 4 bytes of dynamic LDS.
__global__ __launch_bounds__(256) static void k_tfc_to_tcf(const float *__restrict__ in, float *__restrict__ out, size_t NT, int F, int C, int reverse) {
    extern __shared__ __attribute__((aligned(16))) float tile_fc[];         // [f][c + pad]
    const int CP = C + 1;
    const float *src = in + (size_t)blockIdx.x * F * C;
    float *dst = out + (size_t)blockIdx.x * F * C;
    if (!reverse) {
        for (int i = threadIdx.x; i < F * C; i += 256) { const int f = i / C, c = i - f * C; tile_fc[f * CP + c] = src[i]; }
        __syncthreads();
        for (int i = threadIdx.x; i < F * C; i += 256) { const int c = i / F, f = i - c * F; dst[i] = tile_fc[f * CP + c]; }
    } else {
        for (int i = threadIdx.x; i < F * C; i += 256) { const int c = i / F, f = i - c * F; tile_fc[f * CP + c] = src[i]; }
        __syncthreads();
        for (int i = threadIdx.x; i < F * C; i += 256) { const int f = i / C, c = i - f * C; dst[i] = tile_fc[f * CP + c]; }
    }
}

// (the same for rows too large for LDS)
__global__ static void k_tfc_to_tcf_flat(const float *__restrict__ in, float *__restrict__ out, size_t NT, int F, int C, int reverse) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < NT * F * C; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C), f = (int)((i / C) % F);
        const size_t nt = i / ((size_t)C * F), o = (nt * C + c) * F + f;
        if (reverse) out[i] = in[o]; else out[o] = in[i];
    }
}

// ---- torch.optim.AdamW (model.py:283-284) on a flat parameter vector ------------------------------------------------------------------------------
__global__ static void k_adamw_flat(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v, size_t n, float lr,
                                    float b1, float b2, float eps, float wd, float bc1, float bc2) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float w = p[i] * (1.0f - lr * wd);
        const float gi = g[i], mi = b1 * m[i] + (1.0f - b1) * gi, vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        w -= lr / bc1 * mi / (sqrtf(vi) / sqrtf(bc2) + eps);
        p[i] = w;
    }
}
