// Micro-benchmark: one FFN hidden-chunk iteration of the row-chain kernels (rowchain.hip.h) with v_mfma_f32_16x16x32_bf16 (as built) against
// v_mfma_f32_32x32x16_bf16 -- same FLOP, same LDS operand reads, same weight-fragment loads, same SiLU work, 8 waves on one CU:
//   P1: hidden(96 x 32 per wave) = image(96 x 256, LDS) . W1 fragments            (8 k-steps of 32)
//   P2: stream(96 x 32 per wave) += image(96 x 256, LDS) . W2 fragments, with the SiLU of 48 hidden values per lane (5 instructions + convert
//       + an LDS store per pair) spread over its k-steps                        (8 k-steps of 32)
//   one barrier
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/micro/ffn_shape.hip -o tools/micro/ffn_shape && tools/micro/ffn_shape
// Printed: cycles per iteration (matrix pipe alone: 6144 per SIMD).  What it answers: how much of the SiLU's vector time hides behind the
// 32x32x16 shape in THIS loop (the row-chain kernel's iteration takes 10.3 k cycles; DESIGN.md section 4a prices a rewrite on it).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <int SHAPE>
__global__ __launch_bounds__(512) void k(const bf16_t *__restrict__ w, float *out, unsigned long long *cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // image 96 x 256 bf16 (48 KB) + hidden image (48 KB)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 96 * 256 * 2 / 4; i += 512) reinterpret_cast<unsigned *>(smem)[i] = 0x3c003c00u + (i & 255);
    __syncthreads();
    unsigned char *img = smem, *hid = smem + 96 * 512;
    bf16x8 ring[16];
    const bf16_t *wp = w + (size_t)wave * 16 * 512 + lane * 8;
#pragma unroll
    for (int f = 0; f < 16; ++f) ring[f] = *reinterpret_cast<const bf16x8 *>(wp + f * 512);
    f32x4 a1[6][2], xs[6][2];
    f32x16 b1[3], bx[3];
#pragma unroll
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 2; ++j) { a1[i][j] = (f32x4){0, 0, 0, 0}; xs[i][j] = (f32x4){0, 0, 0, 0}; }
#pragma unroll
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 16; ++j) { b1[i][j] = 0.f; bx[i][j] = 0.f; }
    const int r16 = lane & 15, g = lane >> 4, r32 = lane & 31, g2 = lane >> 5;
    auto silu2 = [&](float x0, float x1, unsigned char *dst) {
        f32x2 v = {x0, x1};
        f32x2 e = v * (f32x2){-1.44269504f, -1.44269504f};
        e = (f32x2){__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])} + (f32x2){1.f, 1.f};
        const f32x2 o = v * (f32x2){__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
        *reinterpret_cast<bf16x2 *>(dst) = __builtin_convertvector(o, bf16x2);
    };
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const bf16_t *nxt = wp + (size_t)((it * 2 + 1) & 63) * 8 * 16 * 512;
        // ---- P1
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (SHAPE == 16) {
                bf16x8 a[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) a[i] = *reinterpret_cast<const bf16x8 *>(img + (16 * i + r16) * 512 + (((kk * 4 + g) ^ (r16 & 7)) << 4));
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 6; ++i) a1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[2 * kk + j], a[i], kk ? a1[i][j] : (f32x4){0, 0, 0, 0}, 0, 0, 0);
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    bf16x8 a[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) a[i] = *reinterpret_cast<const bf16x8 *>(img + (32 * i + r32) * 512 + (((kk * 4 + 2 * h + g2) ^ (r32 & 7)) << 4));
#pragma unroll
                    for (int i = 0; i < 3; ++i) b1[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[2 * kk + h], a[i], (kk | h) ? b1[i] : (f32x16){}, 0, 0, 0);
                }
            }
            ring[2 * kk] = *reinterpret_cast<const bf16x8 *>(nxt + (2 * kk) * 512);
            ring[2 * kk + 1] = *reinterpret_cast<const bf16x8 *>(nxt + (2 * kk + 1) * 512);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- P2 with the SiLU of the P1 results folded into its k-steps (6 values per lane and k-step)
        const bf16_t *nx2 = wp + (size_t)((it * 2 + 2) & 63) * 8 * 16 * 512;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (SHAPE == 16) {
                bf16x8 a[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) a[i] = *reinterpret_cast<const bf16x8 *>(hid + (16 * i + r16) * 512 + (((kk * 4 + g) ^ (r16 & 7)) << 4));
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 6; ++i) xs[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[2 * kk + j], a[i], xs[i][j], 0, 0, 0);
                // tiles 3 kk/2 .. : 12 tiles of 4 values over 8 k-steps = 6 values per k-step
                {
                    const int t = (3 * kk) >> 1;                      // tile index 0..11 (two k-steps share the middle tile)
                    const int i = t >> 1, j = t & 1;
                    f32x4 v = a1[i][j];
                    asm volatile("" : "+v"(v));
                    unsigned char *d = hid + 96 * 512 * 0 + (16 * i + r16) * 512 + ((((32 * wave + 16 * j + 4 * g) >> 3) ^ (r16 & 7)) << 4) + (g & 1) * 8;
                    silu2(v[0], v[1], d);
                    silu2(v[2], v[3], d + 4);
                    if (kk & 1) { f32x4 u = a1[(t + 1) >> 1][(t + 1) & 1]; asm volatile("" : "+v"(u)); silu2(u[0], u[1], d + 1024); silu2(u[2], u[3], d + 1028); }
                    else { f32x4 u = a1[(t + 1) >> 1][(t + 1) & 1]; asm volatile("" : "+v"(u)); silu2(u[0], u[1], d + 1024); }
                }
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    bf16x8 a[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) a[i] = *reinterpret_cast<const bf16x8 *>(hid + (32 * i + r32) * 512 + (((kk * 4 + 2 * h + g2) ^ (r32 & 7)) << 4));
#pragma unroll
                    for (int i = 0; i < 3; ++i) bx[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[2 * kk + h], a[i], bx[i], 0, 0, 0);
                    // 48 values per lane over 16 half-steps = 3 per half-step
                    const int e0 = (kk * 2 + h) * 3;                   // element index 0..47 of b1[3][16]
                    float v0 = b1[e0 / 16][e0 % 16], v1 = b1[(e0 + 1) / 16][(e0 + 1) % 16], v2 = b1[(e0 + 2) / 16][(e0 + 2) % 16];
                    asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2));
                    unsigned char *d = hid + (32 * (e0 / 16) + r32) * 512 + ((((32 * wave + 8 * ((e0 % 16) >> 2) + 4 * g2) >> 3) ^ (r32 & 7)) << 4);
                    silu2(v0, v1, d);
                    if (h) silu2(v2, v0, d + 4); else { f32x2 q = {v2, v2}; q = q * (f32x2){-1.44269504f, -1.44269504f}; const float e = __builtin_amdgcn_exp2f(q[0]) + 1.f;
                                                         *reinterpret_cast<bf16_t *>(d + 8) = (bf16_t)(v2 * __builtin_amdgcn_rcpf(e)); }
                }
            }
            ring[2 * kk] = *reinterpret_cast<const bf16x8 *>(nx2 + (2 * kk) * 512);
            ring[2 * kk + 1] = *reinterpret_cast<const bf16x8 *>(nx2 + (2 * kk + 1) * 512);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0.f;
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 2; ++j) r += xs[i][j][0] + a1[i][j][1];
    for (int i = 0; i < 3; ++i) r += bx[i][0] + b1[i][1];
    out[tid] = r;
    if (lane == 0) cyc[wave] = t1 - t0;
}

template <int SHAPE> void run() {
    bf16_t *w; float *out; unsigned long long *cyc;
    hipMalloc(&w, (size_t)66 * 8 * 16 * 512 * 2 + 8 * 16 * 512 * 2); hipMemset(w, 0x3c, (size_t)66 * 8 * 16 * 512 * 2 + 8 * 16 * 512 * 2);
    hipMalloc(&out, 512 * 4); hipHostMalloc(&cyc, 64);
    hipFuncSetAttribute((const void *)k<SHAPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 200;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<SHAPE>, dim3(1), dim3(512), 2 * 96 * 512, 0, w, out, cyc, iters); hipDeviceSynchronize(); }
    unsigned long long mx = 0;
    for (int i = 0; i < 8; ++i) mx = cyc[i] > mx ? cyc[i] : mx;
    printf("%s: %7.0f cycles per hidden-chunk iteration (matrix pipe alone 6144)\n", SHAPE == 16 ? "16x16x32" : "32x32x16", (double)mx / iters);
    hipFree(w); hipFree(out); hipHostFree(cyc);
}
int main() { run<16>(); run<32>(); return 0; }
