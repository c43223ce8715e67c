// LayerNorm over the channel dimension (eps 1e-5, affine) on the fp32 residual stream -- the
// nn.LayerNorm that opens each FFN / attention / conv module (reference feed_forward.py:46,
// attention.py:139,148, convolution.py:136) and closes each block (encoder.py:99).
//
// One wave normalises 4 rows at a time (4 independent load -> reduce -> write chains in flight); a row is
// D/4 float4 chunks, lane owns chunks lane, lane+64, ... (NV = ceil(D/256)); mean and the centred second
// moment (like torch) by wave-shuffle reductions.  Optional chain: a block's closing LayerNorm writes the
// fp32 stream back and the next module's LayerNorm is applied to that result in the same pass (two
// LayerNorms back to back with different affine parameters), writing the GEMM operand in the compute dtype.
// HBM-bound: algorithmic bytes = one fp32 read (+ one fp32 write if chained) + one T write per element.
#pragma once
#include "common.hip.h"

#define COCR_LN_MAX_D 1024

template <typename T, int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float *x, int M, int D, int Dn, float eps,      // Dn: LayerNorm width (columns [Dn, D): zero padding)
                                                        const float *__restrict__ g1, const float *__restrict__ b1,
                                                        float *out_f32,                    // nullable: LN1 result, fp32 (may alias x)
                                                        const float *__restrict__ g2, const float *__restrict__ b2,  // nullable
                                                        T *__restrict__ out_t)             // nullable: LN2(LN1) or LN1 as T
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = (blockIdx.x * 4 + wave) * 4;
    if (row0 >= M) return;
    const float inv_d = 1.0f / (float)Dn;
    const int nchunk = D >> 2, nnorm = Dn >> 2;
    f32x4 ga[NV], ba[NV], gb[NV], bb[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c = lane + 64 * v;
        const bool ok = c < nchunk;
        ga[v] = ok ? *reinterpret_cast<const f32x4 *>(g1 + 4 * c) : (f32x4){0, 0, 0, 0};
        ba[v] = ok ? *reinterpret_cast<const f32x4 *>(b1 + 4 * c) : (f32x4){0, 0, 0, 0};
        if (g2) {
            gb[v] = ok ? *reinterpret_cast<const f32x4 *>(g2 + 4 * c) : (f32x4){0, 0, 0, 0};
            bb[v] = ok ? *reinterpret_cast<const f32x4 *>(b2 + 4 * c) : (f32x4){0, 0, 0, 0};
        }
    }
    f32x4 xv[4][NV];
    float s[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = row0 + r;
        s[r] = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = lane + 64 * v;
            xv[r][v] = (m < M && c < nchunk) ? *reinterpret_cast<const f32x4 *>(x + (size_t)m * D + 4 * c) : (f32x4){0, 0, 0, 0};
            s[r] += xv[r][v][0] + xv[r][v][1] + xv[r][v][2] + xv[r][v][3];
        }
    }
    auto normalise = [&](const f32x4 *gam, const f32x4 *bet) {
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
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float rstd = 1.0f / sqrtf(wave_sum(q[r]) * inv_d + eps);
            s[r] = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xv[r][v][e] = (xv[r][v][e] - mean[r]) * rstd * gam[v][e] + bet[v][e];   // chunks beyond D: gamma = beta = 0
                    s[r] += xv[r][v][e];
                }
        }
    };
    normalise(ga, ba);
    if (out_f32) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = row0 + r;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int c = lane + 64 * v;
                if (m < M && c < nchunk) *reinterpret_cast<f32x4 *>(out_f32 + (size_t)m * D + 4 * c) = xv[r][v];
            }
        }
    }
    if (!out_t) return;
    if (g2) normalise(gb, bb);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = row0 + r;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = lane + 64 * v;
            if (m < M && c < nchunk) {
                T *p = out_t + (size_t)m * D + 4 * c;
                if constexpr (sizeof(T) == 2) { bf16x4 o = {(T)xv[r][v][0], (T)xv[r][v][1], (T)xv[r][v][2], (T)xv[r][v][3]}; *reinterpret_cast<bf16x4 *>(p) = o; }
                else { *reinterpret_cast<f32x4 *>(p) = xv[r][v]; }
            }
        }
    }
}

template <typename T>
static inline void launch_layernorm(hipStream_t s, const float *x, int M, int D, const float *g1, const float *b1,
                                    float *out_f32, const float *g2, const float *b2, T *out_t, int Dn = 0) {
    dim3 grid(ceil_div(M, 16));
    if (Dn <= 0) Dn = D;
    if (D <= 256) hipLaunchKernelGGL((layernorm_kernel<T, 1>), grid, dim3(256), 0, s, x, M, D, Dn, 1e-5f, g1, b1, out_f32, g2, b2, out_t);
    else if (D <= 512) hipLaunchKernelGGL((layernorm_kernel<T, 2>), grid, dim3(256), 0, s, x, M, D, Dn, 1e-5f, g1, b1, out_f32, g2, b2, out_t);
    else hipLaunchKernelGGL((layernorm_kernel<T, 4>), grid, dim3(256), 0, s, x, M, D, Dn, 1e-5f, g1, b1, out_f32, g2, b2, out_t);
}

// x = sum_z partial[z] + bias (split-K partial sums of the frontend output linear), then LayerNorm -> xn: one wave per 4 rows,
// D <= 256 (one float4 chunk per lane).
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_ln_kernel(const float *__restrict__ partial, int splits, size_t zstride, const float *__restrict__ bias,
                                                               int M, int D, int Dn, const float *__restrict__ g1, const float *__restrict__ b1,
                                                               float *__restrict__ x, T *__restrict__ xn) {      // Dn: LayerNorm width (columns [Dn, D) are zero padding)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = (blockIdx.x * 4 + wave) * 4;
    if (row0 >= M) return;
    const int nchunk = D >> 2, nnorm = Dn >> 2, c = lane, cc = min(c, nchunk - 1);
    const float inv_d = 1.0f / (float)Dn;
    const f32x4 bb = *reinterpret_cast<const f32x4 *>(bias + 4 * cc);
    f32x4 v[4];
    float s[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const size_t off = (size_t)min(row0 + r, M - 1) * D + 4 * cc;
        f32x4 t = bb;
        for (int z = 0; z < splits; ++z) t += *reinterpret_cast<const f32x4 *>(partial + z * zstride + off);
        if (c >= nchunk) t = (f32x4){0, 0, 0, 0};
        v[r] = t;
        s[r] = t[0] + t[1] + t[2] + t[3];
        if (row0 + r < M && c < nchunk) *reinterpret_cast<f32x4 *>(x + (size_t)(row0 + r) * D + 4 * c) = t;
    }
    const f32x4 ga = c < nchunk ? *reinterpret_cast<const f32x4 *>(g1 + 4 * cc) : (f32x4){0, 0, 0, 0};
    const f32x4 be = c < nchunk ? *reinterpret_cast<const f32x4 *>(b1 + 4 * cc) : (f32x4){0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float mean = wave_sum(s[r]) * inv_d;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = c < nnorm ? v[r][e] - mean : 0.f; q += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * inv_d + 1e-5f);
        if (row0 + r < M && c < nchunk) {
            T *p = xn + (size_t)(row0 + r) * D + 4 * c;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[r][e] - mean) * rstd * ga[e] + be[e];
            if constexpr (sizeof(T) == 2) { bf16x4 w = {(T)o[0], (T)o[1], (T)o[2], (T)o[3]}; *reinterpret_cast<bf16x4 *>(p) = w; }
            else { *reinterpret_cast<f32x4 *>(p) = (f32x4){o[0], o[1], o[2], o[3]}; }
        }
    }
}
