// Micro-benchmark: cost of "compare, then count" on 64-bit keys for one wave alone on a gfx950 SIMD, by how the compare result
// travels (VCC / SGPR pair) and how far apart producer and consumer are.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/sgpr_dep.hip -o tools/micro/sgpr_dep && tools/micro/sgpr_dep
// Each variant counts, for 8 keys per step, how many are greater than the lane's own key; printed: cycles per key (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>

#define KEYS8 "v"(kj[0]), "v"(kj[1]), "v"(kj[2]), "v"(kj[3]), "v"(kj[4]), "v"(kj[5]), "v"(kj[6]), "v"(kj[7])
template <int V> __global__ __launch_bounds__(64) void k(unsigned *out, unsigned long long *cyc, int iters) {
    unsigned long long kj[8], key = 0x1234567800000000ull + threadIdx.x * 0x01000193u;
    for (int i = 0; i < 8; ++i) kj[i] = 0x1234567000000000ull + (unsigned long long)(i * 977 + threadIdx.x * 13) * 0x100000001ull;
    int rank = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (V == 0) {            // borrow chain through VCC, consumer right behind the producer
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                int tmp;
                asm volatile("v_sub_co_u32 %1, vcc, %2, %4\n\tv_subb_co_u32 %1, vcc, %3, %5, vcc\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc"
                             : "+v"(rank), "=&v"(tmp) : "v"((unsigned)key), "v"((unsigned)(key >> 32)), "v"((unsigned)kj[u]), "v"((unsigned)(kj[u] >> 32)) : "vcc");
            }
        } else if (V == 1) {     // v_cmp_gt_u64 -> VCC -> v_addc, back to back
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("v_cmp_gt_u64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(rank) : "v"(kj[u]), "v"(key) : "vcc");
        } else if (V == 2) {     // eight compares into eight SGPR pairs, then eight add-with-carry
            asm volatile("v_cmp_gt_u64 s[40:41], %1, %9\n\tv_cmp_gt_u64 s[42:43], %2, %9\n\tv_cmp_gt_u64 s[44:45], %3, %9\n\tv_cmp_gt_u64 s[46:47], %4, %9\n\t"
                         "v_cmp_gt_u64 s[48:49], %5, %9\n\tv_cmp_gt_u64 s[50:51], %6, %9\n\tv_cmp_gt_u64 s[52:53], %7, %9\n\tv_cmp_gt_u64 s[54:55], %8, %9\n\t"
                         "v_addc_co_u32 %0, vcc, 0, %0, s[40:41]\n\tv_addc_co_u32 %0, vcc, 0, %0, s[42:43]\n\tv_addc_co_u32 %0, vcc, 0, %0, s[44:45]\n\t"
                         "v_addc_co_u32 %0, vcc, 0, %0, s[46:47]\n\tv_addc_co_u32 %0, vcc, 0, %0, s[48:49]\n\tv_addc_co_u32 %0, vcc, 0, %0, s[50:51]\n\t"
                         "v_addc_co_u32 %0, vcc, 0, %0, s[52:53]\n\tv_addc_co_u32 %0, vcc, 0, %0, s[54:55]"
                         : "+v"(rank) : KEYS8, "v"(key)
                         : "vcc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55");
        } else if (V == 3) {     // the same with 32-bit compares (what a 32-bit key would cost)
            asm volatile("v_cmp_gt_u32 s[40:41], %1, %9\n\tv_cmp_gt_u32 s[42:43], %2, %9\n\tv_cmp_gt_u32 s[44:45], %3, %9\n\tv_cmp_gt_u32 s[46:47], %4, %9\n\t"
                         "v_cmp_gt_u32 s[48:49], %5, %9\n\tv_cmp_gt_u32 s[50:51], %6, %9\n\tv_cmp_gt_u32 s[52:53], %7, %9\n\tv_cmp_gt_u32 s[54:55], %8, %9\n\t"
                         "v_addc_co_u32 %0, vcc, 0, %0, s[40:41]\n\tv_addc_co_u32 %0, vcc, 0, %0, s[42:43]\n\tv_addc_co_u32 %0, vcc, 0, %0, s[44:45]\n\t"
                         "v_addc_co_u32 %0, vcc, 0, %0, s[46:47]\n\tv_addc_co_u32 %0, vcc, 0, %0, s[48:49]\n\tv_addc_co_u32 %0, vcc, 0, %0, s[50:51]\n\t"
                         "v_addc_co_u32 %0, vcc, 0, %0, s[52:53]\n\tv_addc_co_u32 %0, vcc, 0, %0, s[54:55]"
                         : "+v"(rank)
                         : "v"((unsigned)(kj[0] >> 32)), "v"((unsigned)(kj[1] >> 32)), "v"((unsigned)(kj[2] >> 32)), "v"((unsigned)(kj[3] >> 32)),
                           "v"((unsigned)(kj[4] >> 32)), "v"((unsigned)(kj[5] >> 32)), "v"((unsigned)(kj[6] >> 32)), "v"((unsigned)(kj[7] >> 32)), "v"((unsigned)(key >> 32))
                         : "vcc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55");
        } else if (V == 4) {     // eight compares into SGPR pairs, v_cndmask to 0 / 1, three-operand adds
            int t[8];
            asm volatile("v_cmp_gt_u64 s[40:41], %9, %17\n\tv_cmp_gt_u64 s[42:43], %10, %17\n\tv_cmp_gt_u64 s[44:45], %11, %17\n\tv_cmp_gt_u64 s[46:47], %12, %17\n\t"
                         "v_cmp_gt_u64 s[48:49], %13, %17\n\tv_cmp_gt_u64 s[50:51], %14, %17\n\tv_cmp_gt_u64 s[52:53], %15, %17\n\tv_cmp_gt_u64 s[54:55], %16, %17\n\t"
                         "v_cndmask_b32 %1, 0, 1, s[40:41]\n\tv_cndmask_b32 %2, 0, 1, s[42:43]\n\tv_cndmask_b32 %3, 0, 1, s[44:45]\n\tv_cndmask_b32 %4, 0, 1, s[46:47]\n\t"
                         "v_cndmask_b32 %5, 0, 1, s[48:49]\n\tv_cndmask_b32 %6, 0, 1, s[50:51]\n\tv_cndmask_b32 %7, 0, 1, s[52:53]\n\tv_cndmask_b32 %8, 0, 1, s[54:55]\n\t"
                         "v_add3_u32 %0, %0, %1, %2\n\tv_add3_u32 %3, %3, %4, %5\n\tv_add3_u32 %6, %6, %7, %8\n\tv_add3_u32 %0, %0, %3, %6"
                         : "+v"(rank), "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7])
                         : KEYS8, "v"(key)
                         : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55");
        } else if (V == 5) {     // what the compiler makes of the plain expression
#pragma unroll
            for (int u = 0; u < 8; ++u) rank += kj[u] > key;
        } else if (V == 6) {     // a chain of plain dependent v_add_u32 (the floor for 16 dependent VALU instructions)
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %2" : "+v"(rank) : "v"((unsigned)kj[u]), "v"((unsigned)(kj[u] >> 32)));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(kj[u]));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = rank;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int V> void run(const char *what) {
    unsigned *out; unsigned long long *cyc;
    hipMalloc(&out, 64 * 4); hipHostMalloc(&cyc, 8);
    const int iters = 2000;
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, out, cyc, iters); hipDeviceSynchronize(); }
    printf("%-100s %6.1f cycles per key\n", what, (double)cyc[0] / (iters * 8.0));
    hipFree(out); hipHostFree(cyc);
}

int main() {
    run<6>("two dependent v_add_u32 per key (reference)");
    run<0>("v_sub_co / v_subb_co / v_addc_co through VCC, back to back");
    run<1>("v_cmp_gt_u64 vcc + v_addc_co, back to back");
    run<2>("8 x v_cmp_gt_u64 into SGPR pairs, then 8 x v_addc_co");
    run<3>("8 x v_cmp_gt_u32 into SGPR pairs, then 8 x v_addc_co");
    run<4>("8 x v_cmp_gt_u64 into SGPR pairs, 8 x v_cndmask, 4 x v_add3_u32");
    run<5>("rank += kj > key (compiler)");
    return 0;
}
