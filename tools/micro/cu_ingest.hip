// Micro-benchmark: how many bytes per cycle one CU takes in from L2 when every workgroup streams the SAME buffer (the weight stream of the
// row-chain kernels: 8 waves x 16 KiB slices, 1 KiB per wave-instruction, a fixed number of loads in flight per wave).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/cu_ingest.hip -o tools/micro/cu_ingest && tools/micro/cu_ingest
// Variants: V0 global_load_dwordx4 into registers (what the kernels do), V1 LDS-DMA (global_load_lds 16 B per lane) into a ring in LDS,
// V2 = V0 with every workgroup starting at a different slice (staggered: no two CUs ask for the same line at the same moment).
// Printed per (variant, workgroups, loads in flight): bytes per cycle per CU (s_memtime of workgroup 0) and GB/s per CU / chip by events.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(1))) const void *gbl_ptr_t;
typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int V, int DEPTH>
__global__ __launch_bounds__(512) void k(const u32x4 *buf, int slices, int passes, unsigned *out, unsigned long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 ring[DEPTH];
    const int per_wave = slices / 8;                     // 16 KiB slices of this wave per pass
    const int total = per_wave * 16 * passes;            // 1 KiB loads of this wave
    const int start = V == 2 ? (blockIdx.x * 37) % per_wave : 0;
    auto addr = [&](int i) {                             // i-th 1 KiB piece of this wave's stream
        const int sl = ((i >> 4) + start) % per_wave, f = i & 15;
        return buf + ((size_t)(sl * 8 + wave) * 16 + f) * 64 + lane;
    };
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (V == 1) {
        unsigned char *mine = smem + wave * DEPTH * 1024;
        for (int i = 0; i < DEPTH; ++i) __builtin_amdgcn_global_load_lds((gbl_ptr_t)addr(i), (lds_ptr_t)(mine + i * 1024), 16, 0, 0);
        for (int i = DEPTH; i < total; ++i) {
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH - 1) : "memory");
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)addr(i), (lds_ptr_t)(mine + (i % DEPTH) * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc = *reinterpret_cast<u32x4 *>(mine + lane * 16);
    } else {
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) ring[i] = *addr(i);
        for (int i0 = DEPTH; i0 < total; i0 += DEPTH) {
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) {
                acc ^= ring[j];
                ring[j] = *addr(i0 + j);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) acc ^= ring[j];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[threadIdx.x] = 1;
}

template <int V, int DEPTH> void run(const u32x4 *buf, int slices, unsigned *out, unsigned long long *cyc, int wgs) {
    const int passes = 8;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const size_t lds = V == 1 ? 8 * DEPTH * 1024 : 0;
    if (lds > 48 * 1024) hipFuncSetAttribute((const void *)k<V, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k<V, DEPTH><<<wgs, 512, lds>>>(buf, slices, 1, out, cyc);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<V, DEPTH><<<wgs, 512, lds>>>(buf, slices, passes, out, cyc);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)slices * 16384.0 * passes;            // per workgroup
    printf("V%d wgs %3d in flight %2d KiB/wave: %6.1f B/cycle/CU (in-kernel), %6.1f GB/s per CU, %7.2f TB/s chip (events, %.1f us)\n", V, wgs, DEPTH,
           bytes / (double)cyc[0], bytes / (ms * 1e-3) * 1e-9, bytes * wgs / (ms * 1e-3) * 1e-12, ms * 1e3);
    hipEventDestroy(a); hipEventDestroy(b);
}

int main() {
    const int slices = 160;                               // 160 x 16 KiB = 2.5 MiB, the weight bytes of one dominant-kernel launch
    u32x4 *buf; unsigned *out; unsigned long long *cyc;
    hipMalloc((void **)&buf, (size_t)slices * 16384);
    hipMemset(buf, 1, (size_t)slices * 16384);
    hipMalloc((void **)&out, 4096);
    hipHostMalloc((void **)&cyc, 64);
    for (int wgs : {1, 100, 256}) {
        run<0, 8>(buf, slices, out, cyc, wgs);
        run<0, 16>(buf, slices, out, cyc, wgs);
        run<0, 32>(buf, slices, out, cyc, wgs);
        run<2, 16>(buf, slices, out, cyc, wgs);
        run<1, 8>(buf, slices, out, cyc, wgs);
        run<1, 16>(buf, slices, out, cyc, wgs);
    }
    return 0;
}
