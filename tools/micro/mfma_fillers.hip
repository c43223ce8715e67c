// Micro-benchmark: how many vector instructions hide beside bf16 MFMAs on a gfx950 SIMD, by MFMA shape and by waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_fillers.hip -o tools/micro/mfma_fillers && tools/micro/mfma_fillers
// One workgroup on one CU; every wave runs `iters` x 16 "units".  A unit = 32 matrix-pipe cycles of work = two v_mfma_f32_16x16x32_bf16 or
// one v_mfma_f32_32x32x16_bf16 (same FLOP), followed in program order by F filler instructions of a SiLU (x * rcp(1 + exp2(-x log2 e)):
// 3 full-rate + 2 transcendental per value).  sched_barrier(0) pins the order [MFMA(s)][fillers].  Printed: cycles per unit per wave
// (s_memtime), for 4 waves (one per SIMD) and 8 waves (two per SIMD: the matrix pipe is shared, so 64 cycles per unit-pair is the floor).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int VALS> __global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, int iters) {
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(0.01f * (i + threadIdx.x % 7)); fb[i] = (__bf16)(0.02f * i); }
    f32x4 a4[8];
    f32x16 a16[2];
    for (int i = 0; i < 8; ++i) a4[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) a16[i][j] = 0.f;
    float v[4] = {0.3f + threadIdx.x * 1e-3f, -0.2f, 0.7f, 1.1f}, o[4] = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (SHAPE == 16) {
                a4[(2 * u) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, a4[(2 * u) & 7], 0, 0, 0);
                a4[(2 * u + 1) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, a4[(2 * u + 1) & 7], 0, 0, 0);
            } else {
                a16[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, a16[u & 1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < VALS; ++q) {          // one SiLU value = 5 instructions (mul, exp2, add, rcp, mul)
                float e = __builtin_amdgcn_exp2f(v[q] * -1.4426950408889634f) + 1.0f;
                o[q] += v[q] * __builtin_amdgcn_rcpf(e);
                asm volatile("" : "+v"(v[q]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = o[0] + o[1] + o[2] + o[3];
    for (int i = 0; i < 8; ++i) r += a4[i][0];
    for (int i = 0; i < 2; ++i) r += a16[i][0];
    out[threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int SHAPE, int VALS> void run() {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 1024 * 4); hipHostMalloc(&cyc, 16 * 8);
    const int iters = 500;
    for (int waves : {4, 8}) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<SHAPE, VALS>), dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters); hipDeviceSynchronize(); }
        unsigned long long mx = 0;
        for (int i = 0; i < waves; ++i) mx = cyc[i] > mx ? cyc[i] : mx;
        printf("%s  %d SiLU value(s) (%2d instructions) per unit  %d waves/SIMD: %6.1f cycles per unit per wave  (%.1f per SIMD-unit; floor 32)\n",
               SHAPE == 16 ? "2 x 16x16x32" : "1 x 32x32x16", VALS, 5 * VALS, waves / 4, (double)mx / (iters * 16.0), (double)mx / (iters * 16.0) / (waves / 4));
    }
    hipFree(out); hipHostFree(cyc);
}
int main() {
    run<16, 0>(); run<32, 0>();
    run<16, 1>(); run<32, 1>();
    run<16, 2>(); run<32, 2>();
    run<16, 3>(); run<32, 3>();
    return 0;
}
