// Micro-benchmark: WHICH vector instructions hide beside v_mfma_f32_16x16x32_bf16 on a gfx950 SIMD, and in which program order.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_overlap.hip -o tools/micro/mfma_overlap && tools/micro/mfma_overlap
// One workgroup on one CU, 4 or 8 waves (1 or 2 per SIMD).  A unit = two MFMAs (32 matrix-pipe cycles) + fillers worth ~28 issue cycles:
//   KIND 0: 7 x v_fma_f32 (full rate)      KIND 1: 3 x v_exp_f32 + 1 x v_fma (transcendentals)      KIND 2: 7 x v_pk_fma_f32
//   ORDER 0: [MFMA MFMA][fillers]          ORDER 1: [MFMA][half the fillers][MFMA][the other half]
// Printed: cycles per unit per SIMD (floor 32 = the matrix pipe alone; 60 = no overlap at all).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int ORDER, int NF> __global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, int iters) {
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(0.01f * (i + threadIdx.x % 7)); fb[i] = (__bf16)(0.02f * i); }
    f32x4 a4[8];
    for (int i = 0; i < 8; ++i) a4[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float v[8];
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) { v[i] = 0.3f + threadIdx.x * 1e-3f + i; p[i] = (f32x2){v[i], -v[i]}; }
    auto filler = [&](int q) {
        if (KIND == 0) { v[q & 7] = __builtin_fmaf(v[q & 7], 1.0001f, 0.5f); }
        else if (KIND == 1) { if (q % 4 == 3) v[q & 7] = __builtin_fmaf(v[q & 7], 1.0001f, 0.5f); else v[q & 7] = __builtin_amdgcn_exp2f(v[q & 7]); }
        else { p[q & 7] = __builtin_elementwise_fma(p[q & 7], (f32x2){1.0001f, 0.9999f}, (f32x2){0.5f, 0.25f}); }
        asm volatile("" : "+v"(v[q & 7]), "+v"(p[q & 7]));
    };
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            a4[(2 * u) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, a4[(2 * u) & 7], 0, 0, 0);
            if (ORDER == 1) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < NF / 2; ++q) filler(q);
                __builtin_amdgcn_sched_barrier(0);
            }
            a4[(2 * u + 1) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, a4[(2 * u + 1) & 7], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = (ORDER == 1 ? NF / 2 : 0); q < NF; ++q) filler(q);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0.f;
    for (int i = 0; i < 8; ++i) r += a4[i][0] + v[i] + p[i][0];
    out[threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int KIND, int ORDER, int NF> void run(const char *what) {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 1024 * 4); hipHostMalloc(&cyc, 16 * 8);
    const int iters = 500;
    for (int waves : {4, 8}) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<KIND, ORDER, NF>), dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters); hipDeviceSynchronize(); }
        unsigned long long mx = 0;
        for (int i = 0; i < waves; ++i) mx = cyc[i] > mx ? cyc[i] : mx;
        printf("%-34s order %d  %d wave(s)/SIMD: %6.1f cycles per unit per SIMD\n", what, ORDER, waves / 4, (double)mx / (iters * 16.0) / (waves / 4));
    }
    hipFree(out); hipHostFree(cyc);
}
int main() {
    run<0, 0, 0>("no fillers");
    run<0, 0, 7>("7 x v_fma_f32"); run<0, 1, 7>("7 x v_fma_f32");
    run<1, 0, 4>("3 x v_exp_f32 + v_fma"); run<1, 1, 4>("3 x v_exp_f32 + v_fma");
    run<2, 0, 7>("7 x v_pk_fma_f32"); run<2, 1, 7>("7 x v_pk_fma_f32");
    run<0, 0, 14>("14 x v_fma_f32"); run<0, 1, 14>("14 x v_fma_f32");
    return 0;
}
