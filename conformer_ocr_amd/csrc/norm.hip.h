// LayerNorm over the channel dimension (eps 1e-5, affine) on the fp32 residual stream -- the
// nn.LayerNorm that opens each FFN / attention / conv module (reference feed_forward.py:46,
// attention.py:139,148, convolution.py:136) and closes each block (encoder.py:99).
//
// One wave per row, the row held in registers (D <= 1024), two wave-shuffle reductions (mean, then
// the centred second moment, like torch).  Optional chain: a block's closing LayerNorm writes the
// fp32 stream back in place and the next module's LayerNorm is applied to that result in the same
// pass (two LayerNorms back to back with different affine parameters), writing the GEMM operand in
// the compute dtype.  HBM-bound: algorithmic bytes = one fp32 read (+ one fp32 write if chained)
// + one T write per element.
#pragma once
#include "common.hip.h"

#define COCR_LN_MAX_PER_LANE 16   // D <= 64 * 16

template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ x, int M, int D, float eps,
                                                        const float *__restrict__ g1, const float *__restrict__ b1,
                                                        float *__restrict__ out_f32,      // nullable: LN1 result, fp32
                                                        const float *__restrict__ g2, const float *__restrict__ b2,  // nullable
                                                        T *__restrict__ out_t)            // nullable: LN2(LN1) or LN1 as T
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float *xr = x + (size_t)row * D;
    float v[COCR_LN_MAX_PER_LANE];
    const float inv_d = 1.0f / (float)D;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < COCR_LN_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < D ? xr[c] : 0.f;
        s += v[i];
    }
    float mean = wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < COCR_LN_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        const float d = c < D ? v[i] - mean : 0.f;
        q += d * d;
    }
    float rstd = 1.0f / sqrtf(wave_sum(q) * inv_d + eps);
#pragma unroll
    for (int i = 0; i < COCR_LN_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        if (c < D) v[i] = (v[i] - mean) * rstd * g1[c] + b1[c];
    }
    if (out_f32) {
#pragma unroll
        for (int i = 0; i < COCR_LN_MAX_PER_LANE; ++i) {
            const int c = lane + 64 * i;
            if (c < D) out_f32[(size_t)row * D + c] = v[i];
        }
    }
    if (!out_t) return;
    if (g2) {
        s = 0.f;
#pragma unroll
        for (int i = 0; i < COCR_LN_MAX_PER_LANE; ++i) s += (lane + 64 * i < D) ? v[i] : 0.f;
        mean = wave_sum(s) * inv_d;
        q = 0.f;
#pragma unroll
        for (int i = 0; i < COCR_LN_MAX_PER_LANE; ++i) {
            const float d = (lane + 64 * i < D) ? v[i] - mean : 0.f;
            q += d * d;
        }
        rstd = 1.0f / sqrtf(wave_sum(q) * inv_d + eps);
#pragma unroll
        for (int i = 0; i < COCR_LN_MAX_PER_LANE; ++i) {
            const int c = lane + 64 * i;
            if (c < D) v[i] = (v[i] - mean) * rstd * g2[c] + b2[c];
        }
    }
#pragma unroll
    for (int i = 0; i < COCR_LN_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        if (c < D) out_t[(size_t)row * D + c] = from_f32<T>(v[i]);
    }
}

template <typename T>
static inline void launch_layernorm(hipStream_t s, const float *x, int M, int D, const float *g1, const float *b1,
                                    float *out_f32, const float *g2, const float *b2, T *out_t) {
    hipLaunchKernelGGL((layernorm_kernel<T>), dim3(ceil_div(M, 4)), dim3(256), 0, s, x, M, D, 1e-5f, g1, b1, out_f32, g2, b2, out_t);
}
