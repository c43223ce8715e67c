// Micro-benchmark: latency of a workgroup's FIRST loads after a kernel boundary (all workgroups start together), by what the buffer is:
// written by the kernel launched just before on the same stream (the activations of the next launch), or read-only and read by the launch
// before as well (the weights).  One 16-byte load per lane (1 KiB per wave), 8 waves per workgroup, every workgroup its own rows.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/first_touch.hip -o tools/micro/first_touch && tools/micro/first_touch
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void writer(u32x4 *buf, int n16) {
    for (size_t i = blockIdx.x * 512 + threadIdx.x; i < (size_t)n16; i += (size_t)gridDim.x * 512) buf[i] = (u32x4){(unsigned)i, 1, 2, 3};
}
// each wave: `nload` loads of 1 KiB, `shared` = every workgroup reads the same addresses
__global__ __launch_bounds__(512) void reader(const u32x4 *buf, int nload, int shared, unsigned *out, unsigned long long *cyc) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32x4 *p = buf + ((size_t)(shared ? 0 : blockIdx.x) * 8 + wave) * 64 * 64 + lane;      // 64 KiB apart per wave
    u32x4 v[16];
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 16; ++i) if (i < nload) v[i] = p[i * 64];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) if (i < nload) acc ^= v[i][0] ^ v[i][3];
    if (lane == 0 && blockIdx.x == 7) cyc[wave] = t1 - t0;
    if (acc == 0x1234567u) out[0] = acc;
}

int main() {
    const size_t bytes = 256ull * 8 * 64 * 1024;           // 128 MiB
    u32x4 *buf; unsigned *out; unsigned long long *cyc;
    (void)hipMalloc((void **)&buf, bytes);
    (void)hipMalloc((void **)&out, 64);
    (void)hipHostMalloc((void **)&cyc, 64);
    (void)hipMemset(buf, 0, bytes);
    for (int nload : {1, 8, 16}) {
        for (int shared : {0, 1}) {
            for (int mode = 0; mode < 3; ++mode) {       // 0: after a writer of the buffer, 1: after a reader of the same data, 2: after an idle gap
                for (int rep = 0; rep < 2; ++rep) {
                    if (mode == 0) writer<<<1024, 512>>>(buf, (int)(100 * 8 * 64 * 1024 / 16));
                    if (mode == 1) reader<<<100, 512>>>(buf, nload, shared, out, cyc);
                    if (mode == 2) (void)hipDeviceSynchronize();
                    reader<<<100, 512>>>(buf, nload, shared, out, cyc);
                    (void)hipDeviceSynchronize();
                }
                printf("%2d KiB per wave, %s, %s: first loads back after %5llu / %5llu / %5llu cycles (waves 0, 3, 7 of workgroup 7)\n", nload,
                       shared ? "all workgroups the same rows" : "every workgroup its own rows",
                       mode == 0 ? "buffer written by the launch before" : mode == 1 ? "read by the launch before            " : "after an idle gap                   ",
                       cyc[0], cyc[3], cyc[7]);
            }
        }
    }
    return 0;
}
