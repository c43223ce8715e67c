// Weight re-layout kernels run once per model (cocr_api.hip: ensure_packed).
#pragma once
#include "common.hip.h"

// fragment-major copy of a row-major (N, K) bf16 matrix, optionally scaled (a power of two: exact)
__global__ __launch_bounds__(256) void pack_frag_kernel(const bf16_t *__restrict__ src, bf16_t *__restrict__ dst, int N, int K, float scale) {
    const size_t units = (size_t)N * K / 8;
    for (size_t u = (size_t)blockIdx.x * 256 + threadIdx.x; u < units; u += (size_t)gridDim.x * 256) {
        const int lane = (int)(u & 63), j = (int)((u >> 6) & 1);
        const size_t blk = u >> 7;                        // (pair, kt)
        const int kt = (int)(blk % (K / 32)), pair = (int)(blk / (K / 32));
        const int n = pair * 32 + j * 16 + (lane & 15), k = kt * 32 + 8 * (lane >> 4);
        bf16x8 v = *reinterpret_cast<const bf16x8 *>(src + (size_t)n * K + k);
        if (scale != 1.0f) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] * scale);
        }
        *reinterpret_cast<bf16x8 *>(dst + u * 8) = v;
    }
}

