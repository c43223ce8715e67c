// Micro-benchmark: issue cost (cycles per wave-instruction) of the VALU forms a depthwise convolution could use on gfx950.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rates.hip -o tools/micro/valu_rates && tools/micro/valu_rates
// One workgroup of W waves on one CU, each wave runs N dependent-free instructions (8 independent accumulator chains), timed with
// s_memtime; printed: cycles per instruction per wave at 4 waves (1 per SIMD) and 8 waves (2 per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

template <int MODE> __global__ void k(float *out, unsigned long long *cyc, int iters) {
    f32x2 a[8];
    float s[8];
    for (int i = 0; i < 8; ++i) { a[i] = (f32x2){(float)threadIdx.x + i, 1.0f}; s[i] = threadIdx.x + i; }
    f32x2 w = {1.0001f, 0.9999f}, x = {0.5f, 0.25f};
    f16x2 hw = {(_Float16)1.001f, (_Float16)0.999f}, hx = {(_Float16)0.5f, (_Float16)0.25f};
    bf16x2 bw = {(__bf16)1.001f, (__bf16)0.999f}, bx = {(__bf16)0.5f, (__bf16)0.25f};
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) a[i] = __builtin_elementwise_fma(w, x, a[i]);                    // v_pk_fma_f32
                if (MODE == 1) s[i] = __builtin_fmaf(w[0], x[0], s[i]);                           // v_fma_f32 / v_fmac
                if (MODE == 2) s[i] = __builtin_amdgcn_fdot2(hw, hx, s[i], false);                // v_dot2_f32_f16
                if (MODE == 3) s[i] = __builtin_amdgcn_fdot2_f32_bf16(bw, bx, s[i], false);       // v_dot2_f32_bf16 (if the target has it)
                if (MODE == 4) s[i] = __builtin_amdgcn_exp2f(s[i]);                               // v_exp_f32
                if (MODE == 5) s[i] = __builtin_amdgcn_rcpf(s[i]);                                // v_rcp_f32
                if (MODE == 6) s[i] = __builtin_amdgcn_rsqf(s[i]);                                // v_rsq_f32
                asm volatile("" : "+v"(x), "+v"(hx), "+v"(bx));
            }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0;
    for (int i = 0; i < 8; ++i) r += a[i][0] + a[i][1] + s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

// Do the matrix pipe and the VALU overlap?  Waves 0..3 (one per SIMD) issue dependent-free MFMAs, waves 4..7 (their SIMD partners)
// issue v_pk_fma_f32; printed: cycles of each group alone and together.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void kco(float *out, unsigned long long *cyc, int iters, int mode) {      // mode 1: MFMA waves only, 2: VALU waves only, 3: both
    const int wave = threadIdx.x >> 6;
    const bool mf = wave < 4;
    f32x4 acc[4];
    f32x2 a[8];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 8; ++i) a[i] = (f32x2){(float)threadIdx.x + i, 1.0f};
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(0.01f * i); fb[i] = (__bf16)(0.02f * i); }
    f32x2 w = {1.0001f, 0.9999f}, x = {0.5f, 0.25f};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (mf && (mode & 1)) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i], 0, 0, 0);
    }
    if (!mf && (mode & 2)) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) { a[i] = __builtin_elementwise_fma(w, x, a[i]); asm volatile("" : "+v"(x)); }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0;
    for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][3];
    for (int i = 0; i < 8; ++i) r += a[i][0] + a[i][1];
    out[threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}
void coexec() {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 1024 * 4); hipHostMalloc(&cyc, 16 * 8);
    const int iters = 2000;
    for (int mode = 1; mode <= 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kco, dim3(1), dim3(512), 0, 0, out, cyc, iters, mode); hipDeviceSynchronize(); }
        printf("co-issue mode %d (1 = 32 MFMA 16x16x32 per iteration on waves 0-3, 2 = 32 v_pk_fma_f32 on waves 4-7, 3 = both): MFMA wave %.1f cycles/iteration, "
               "VALU wave %.1f cycles/iteration\n", mode, (double)cyc[0] / iters, (double)cyc[4] / iters);
    }
}

template <int MODE> void run(const char *name) {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 1024 * 4); hipHostMalloc(&cyc, 16 * 8);
    const int iters = 2000;
    for (int waves : {4, 8, 16}) {
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        unsigned long long mx = 0;
        for (int i = 0; i < waves; ++i) mx = cyc[i] > mx ? cyc[i] : mx;
        printf("%-18s %2d waves/CU: %.2f cycles per wave-instruction (per SIMD: %.2f per instruction issued)\n", name, waves, (double)mx / (iters * 32.0),
               (double)mx / (iters * 32.0) / (waves / 4.0));
    }
}
int main() {
    run<0>("v_pk_fma_f32");
    run<1>("v_fma_f32");
    run<2>("v_dot2_f32_f16");
    run<3>("v_dot2_f32_bf16");
    run<4>("v_exp_f32");
    run<5>("v_rcp_f32");
    run<6>("v_rsq_f32");
    coexec();
    return 0;
}
