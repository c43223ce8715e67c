// Micro-benchmark: what straight-line code costs when it does not come from the instruction cache.  The row-chain kernels are ~90 KB of
// unrolled code that a workgroup executes once (only the FFN chunk loops repeat, three times); the instruction cache holds 64 KB.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/ifetch.hip -o tools/micro/ifetch && tools/micro/ifetch
// Each kernel executes N 8-byte vector instructions (v_add3_u32, independent enough to issue every 4 cycles) either as one straight run
// (`.rept N`) or as a loop over a 2 KB body; printed: cycles per instruction for one wave of workgroup 0 (s_memtime), cold launch and
// the launch right after, by waves per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>

#define BODY(N) asm volatile(".rept " #N "\n\tv_add3_u32 %0, %0, %1, %2\n\tv_add3_u32 %1, %1, %0, %2\n\t.endr" : "+v"(a), "+v"(b) : "v"(c));

template <int KB> __global__ __launch_bounds__(512) void straight(unsigned *out, unsigned long long *cyc) {
    unsigned a = threadIdx.x, b = blockIdx.x, c = 3;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (KB == 16) BODY(1024) else if (KB == 48) BODY(3072) else if (KB == 96) BODY(6144) else BODY(12288)
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (a + b == 0x1234567u) out[0] = a;
}
template <int KB> __global__ __launch_bounds__(512) void looped(unsigned *out, unsigned long long *cyc) {
    unsigned a = threadIdx.x, b = blockIdx.x, c = 3;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < KB / 2; ++i) BODY(128)
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (a + b == 0x1234567u) out[0] = a;
}

template <typename F> void run(const char *name, F kern, int kb, int wgs, int threads, unsigned *out, unsigned long long *cyc) {
    double r[2];
    for (int rep = 0; rep < 2; ++rep) {
        kern<<<wgs, threads>>>(out, cyc);
        (void)hipDeviceSynchronize();
        r[rep] = (double)cyc[0] / (kb * 128.0);
    }
    printf("%-8s %3d KB, %3d workgroups x %d waves: %6.2f cycles per instruction (first launch), %6.2f (second)\n", name, kb, wgs, threads / 64, r[0], r[1]);
}

int main() {
    unsigned *out; unsigned long long *cyc;
    (void)hipMalloc((void **)&out, 64);
    (void)hipHostMalloc((void **)&cyc, 64);
    for (int threads : {64, 512}) {
        for (int wgs : {1, 100}) {
            run("straight", straight<16>, 16, wgs, threads, out, cyc);
            run("straight", straight<48>, 48, wgs, threads, out, cyc);
            run("straight", straight<96>, 96, wgs, threads, out, cyc);
            run("straight", straight<192>, 192, wgs, threads, out, cyc);
            run("looped", looped<96>, 96, wgs, threads, out, cyc);
        }
    }
    return 0;
}
