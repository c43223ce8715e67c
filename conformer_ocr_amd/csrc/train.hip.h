// Training step of the output layer -- the tail of the reference's `training_step` (model.py:147-152: `loss.backward()` through
// `self.nn['decoder'] = nn.Linear(encoder_dim, num_classes)`, model.py:115,127,135) and the optimizer update the reference configures
// by default (`torch.optim.AdamW`, model.py:47,283-284; default_specs.py:27-30).  The encoder's backward is not part of this library
// yet: these kernels cover the layer between the encoder output and the criterion (ctc_loss.hip.h), i.e. the reference's training
// with the backbone frozen (`freeze_backbone`, default_specs.py:47, cli/train.py:154-155).
//
//   probits[m, c] = sum_d y[m, d] W[c, d] + b[c]            (m = line * T + frame; y = the encoder output the forward multiplied)
//   dW[c, d] = sum_m g[m, c] y[m, d]      db[c] = sum_m g[m, c]      dy[m, d] = sum_c g[m, c] W[c, d]      (g = d loss / d probits)
//
// dW / db reduce over the M = N T rows: row chunks of 128 -> fp32 partial sums -> one reduction pass in chunk order: deterministic,
// no floating-point atomics.  fp32 FMAs on the VALU (0.5 GFLOP at the bench shape: not worth an MFMA layout), y and W read in the
// forward's compute dtype (the gradient of the function the forward actually computed).  HBM-bound: algorithmic bytes per launch
// = M C 4 (g) + M D sizeof(T) (y) + (M / 128) C D 4 (partials written, then read once).
#pragma once
#include "common.hip.h"

#define COCR_TR_ROWS 128        // rows per workgroup of the weight-gradient kernel
#define COCR_TR_CT 32           // classes per workgroup

template <typename T>
__global__ __launch_bounds__(256) void decoder_wgrad_kernel(const float *__restrict__ g, const T *__restrict__ y, int M, int C, int D,
                                                            float *__restrict__ part_w, float *__restrict__ part_b) {
    __shared__ __attribute__((aligned(16))) float gs[COCR_TR_ROWS][COCR_TR_CT];
    const int tid = threadIdx.x, chunk = blockIdx.x, c0 = blockIdx.y * COCR_TR_CT, r0 = chunk * COCR_TR_ROWS;
    for (int i = tid; i < COCR_TR_ROWS * COCR_TR_CT; i += 256) {
        const int r = i / COCR_TR_CT, c = i - r * COCR_TR_CT;
        gs[r][c] = (r0 + r < M && c0 + c < C) ? g[(size_t)(r0 + r) * C + c0 + c] : 0.f;
    }
    __syncthreads();
    const int rows = min(COCR_TR_ROWS, M - r0);
    for (int d = tid; d < D; d += 256) {
        float acc[COCR_TR_CT];
#pragma unroll
        for (int c = 0; c < COCR_TR_CT; ++c) acc[c] = 0.f;
        for (int r = 0; r < rows; ++r) {
            const float yv = to_f32(y[(size_t)(r0 + r) * D + d]);
#pragma unroll
            for (int c4 = 0; c4 < COCR_TR_CT / 4; ++c4) {
                const f32x4 gv = *reinterpret_cast<const f32x4 *>(&gs[r][4 * c4]);      // same address in every lane: broadcast
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[4 * c4 + e] = fmaf(gv[e], yv, acc[4 * c4 + e]);
            }
        }
#pragma unroll
        for (int c = 0; c < COCR_TR_CT; ++c)
            if (c0 + c < C) part_w[((size_t)chunk * C + c0 + c) * D + d] = acc[c];
    }
    if (tid < COCR_TR_CT && c0 + tid < C) {
        float sum = 0.f;
        for (int r = 0; r < rows; ++r) sum += gs[r][tid];
        part_b[(size_t)chunk * C + c0 + tid] = sum;
    }
}

// out[i] = sum over chunks (ascending) of part[chunk][i]
__global__ __launch_bounds__(256) void chunk_reduce_kernel(const float *__restrict__ part, int chunks, size_t n, float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < chunks; ++k) s += part[(size_t)k * n + i];
    out[i] = s;
}

// dy[m, d] = sum_c g[m, c] W[c, d]: 16 rows per workgroup, thread = column d, classes in tiles of 128 through LDS
template <typename T>
__global__ __launch_bounds__(256) void decoder_igrad_kernel(const float *__restrict__ g, const T *__restrict__ W, int M, int C, int D,
                                                            float *__restrict__ dy) {
    __shared__ __attribute__((aligned(16))) float gs[128][16];          // [class][row]: a class's 16 row values are one 64-byte broadcast read
    const int tid = threadIdx.x, r0 = blockIdx.x * 16;
    for (int d0 = 0; d0 < D; d0 += 256) {
        const int d = d0 + tid;
        float acc[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int c0 = 0; c0 < C; c0 += 128) {
            __syncthreads();
            for (int i = tid; i < 128 * 16; i += 256) {
                const int r = i >> 7, c = i & 127;                       // consecutive threads: consecutive classes of one row (coalesced)
                gs[c][r] = (r0 + r < M && c0 + c < C) ? g[(size_t)(r0 + r) * C + c0 + c] : 0.f;
            }
            __syncthreads();
            const int nc = min(128, C - c0);
            if (d < D)
                for (int c = 0; c < nc; ++c) {
                    const float w = to_f32(W[(size_t)(c0 + c) * D + d]);
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x4 gv = *reinterpret_cast<const f32x4 *>(&gs[c][4 * r4]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[4 * r4 + e] = fmaf(gv[e], w, acc[4 * r4 + e]);
                    }
                }
        }
        if (d < D)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (r0 + r < M) dy[(size_t)(r0 + r) * D + d] = acc[r];
    }
}

// torch.optim.AdamW (decoupled weight decay, no amsgrad), one step over n parameters:
//   p *= 1 - lr wd;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// The fp32 master copy stays in `p`; `blob_t` / `blob_f32` receive the value the forward reads (compute dtype / fp32), whichever is given.
template <typename T>
__global__ __launch_bounds__(256) void adamw_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m1, float *__restrict__ m2, size_t n,
                                                    float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt, T *__restrict__ blob_t,
                                                    float *__restrict__ blob_f32) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = b1 * m1[i] + (1.0f - b1) * gi;          // torch: exp_avg.lerp_(grad, 1 - beta1)
    const float vi = b2 * m2[i] + (1.0f - b2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi;
    m1[i] = mi;
    m2[i] = vi;
    if (blob_t) blob_t[i] = from_f32<T>(pi);
    if (blob_f32) blob_f32[i] = pi;
}
