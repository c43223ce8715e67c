// Micro-benchmark (round 4): the weight stream of a SMALL batch.  At B = 1 a row-chain launch is 10 workgroups on 10 CUs of 8 different XCDs, each
// streaming the layer's 2.5 MiB of weights once: nothing has put them into that XCD's L2 (a forward touches 43 MB of weights between two uses of
// a layer), so every line comes from the Infinity Cache / HBM and the stream runs at (bytes in flight) / (that latency).  Variants:
//   warm : the same 2.5 MiB every pass (L2-resident after the first pass): what tools/micro/cu_ingest measures
//   cold : pass p streams region p of a 160 MiB buffer (never in L2; in the Infinity Cache from the second sweep on)
//   pref : cold, but a second kernel on another stream has touched region p (8 workgroups x 4: one set per XCD) while pass p - 1 ran
//   hipcc --offload-arch=gfx950 -O3 tools/micro/cold_ingest.hip -o tools/micro/cold_ingest && tools/micro/cold_ingest
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int DEPTH = 16;

__global__ __launch_bounds__(512) void stream_k(const u32x4 *buf, int slices, unsigned *out, unsigned long long *cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4 acc = {0, 0, 0, 0}, ring[DEPTH];
    const int total = (slices / 8) * 16;
    auto addr = [&](int i) { return buf + ((size_t)((i >> 4) * 8 + wave) * 16 + (i & 15)) * 64 + lane; };
    const unsigned long long t0 = __builtin_readcyclecounter();
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
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[threadIdx.x] = 1;
}

// touch `bytes` at `buf`: workgroup b of nb covers the 128-byte lines b, b + nb, ... of its XCD's share; every XCD (b % 8) reads ALL lines
__global__ __launch_bounds__(256) void prefetch_k(const unsigned char *buf, size_t bytes, unsigned *out) {
    const int xcd_wgs = gridDim.x / 8, mine = blockIdx.x / 8;        // workgroups per XCD, this one's index inside its XCD
    const size_t lines = bytes / 128;
    unsigned acc = 0;
    for (size_t l = (size_t)mine * 256 + threadIdx.x; l < lines; l += (size_t)xcd_wgs * 256)
        acc ^= *reinterpret_cast<const unsigned *>(buf + l * 128);
    if (acc == 0x12345u) out[threadIdx.x] = 1;
}

int main() {
    const int slices = 160;
    const size_t region = (size_t)slices * 16384, nreg = 64;
    u32x4 *buf; unsigned *out; unsigned long long *cyc;
    hipMalloc((void **)&buf, region * nreg);
    hipMemset(buf, 1, region * nreg);
    hipMalloc((void **)&out, 4096);
    hipHostMalloc((void **)&cyc, 64);
    hipStream_t s1, s2;
    hipStreamCreate(&s1); hipStreamCreate(&s2);
    hipEvent_t ev[nreg + 1], a, b;
    for (auto &e : ev) hipEventCreate(&e);
    hipEventCreate(&a); hipEventCreate(&b);
    for (int wgs : {1, 10, 40}) {
        for (int variant = 0; variant < 3; ++variant) {
            double cycles = 0;
            float ms = 0;
            for (int sweep = 0; sweep < 2; ++sweep) {             // second sweep: the buffer has been through the Infinity Cache once
                cycles = 0;
                hipDeviceSynchronize();
                hipEventRecord(a, s1);
                for (size_t p = 0; p < nreg; ++p) {
                    const u32x4 *src = variant == 0 ? buf : buf + p * (region / 16);
                    if (variant == 2) {
                        // region p + 1 is touched while pass p runs (the prefetch waits for pass p - 1, like a graph branch forked behind it)
                        if (p == 0) { prefetch_k<<<32, 256, 0, s2>>>((const unsigned char *)src, region, out); hipEventRecord(ev[0], s2); hipStreamWaitEvent(s1, ev[0], 0); }
                    }
                    stream_k<<<wgs, 512, 0, s1>>>(src, slices, out, cyc);
                    if (variant == 2 && p + 1 < nreg) {
                        prefetch_k<<<32, 256, 0, s2>>>((const unsigned char *)(buf + (p + 1) * (region / 16)), region, out);
                        hipEventRecord(ev[p + 1], s2);
                        hipStreamWaitEvent(s1, ev[p + 1], 0);     // (the next pass starts when both are done: the prefetch is the shorter one)
                    }
                }
                hipEventRecord(b, s1);
                hipDeviceSynchronize();
                hipEventElapsedTime(&ms, a, b);
                cycles = (double)cyc[0];
            }
            printf("%s wgs %2d: last pass %6.1f B/cycle/CU in-kernel; %6.1f us per pass by events (%d passes)\n", variant == 0 ? "warm" : variant == 1 ? "cold" : "pref", wgs,
                   (double)region / cycles, ms * 1e3 / nreg, (int)nreg);
        }
    }
    return 0;
}
