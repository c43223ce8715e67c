// CTC decoding on the device -- replaces the per-line host loop of the reference
// (pred.py:137-145,157-164,172-178; model.py:163-168: D2H of all logits, then
// kraken.lib.ctc_decoder.greedy_decoder per line in Python).  Only the compact label records
// leave the GPU.  Integer outputs (labels, starts, ends, counts) are exact; conf is the max over a
// run of the winning logit (greedy) / softmax probability (beam).
#pragma once
#include "common.hip.h"

// ---- greedy: argmax per frame (first index on ties, like numpy), merge runs, drop blank ---------
// Per frame: lanes stride the classes (coalesced), keep (value, index) with strict > so the lowest index wins
// inside a lane, then a butterfly that prefers the larger value and, on equality, the lower index.
// HBM-bound: algorithmic bytes = N*T*ncls*4 read.
// Two launches: (1) one wave per frame over the whole batch (N*T waves: fills the chip) writes the frame's
// argmax label and max logit; (2) one wave per line scans the frame labels in chunks of 64: a frame opens a run
// if its label differs from its predecessor's; ballot + popcount give each non-blank run its output slot; the
// run's end and confidence are found by walking forward.
__global__ __launch_bounds__(256) void ctc_argmax_kernel(const float *__restrict__ logits, int T, int ncls, int frames,
                                                         const int32_t *__restrict__ lens,
                                                         int32_t *__restrict__ flab, float *__restrict__ fval) {
    const int lane = threadIdx.x & 63;
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6);          // frame index n * T + t
    if (f >= frames) return;
    const int n = f / T, t = f - n * T;
    if (t >= lens[n]) return;                                  // frames beyond the line's length are never read
    const float *lg = logits + (size_t)f * ncls;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < ncls; c += 64) {
        const float v = lg[c];
        if (v > best || bi == 0x7fffffff) { best = v; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) { flab[f] = bi; fval[f] = best; }
}

__global__ __launch_bounds__(64) void ctc_collapse_kernel(int T, const int32_t *__restrict__ lens, const int32_t *__restrict__ flab_all,
                                                          const float *__restrict__ fval_all,
                                                          int32_t *__restrict__ labels, int32_t *__restrict__ starts,
                                                          int32_t *__restrict__ ends, float *__restrict__ conf,
                                                          int32_t *__restrict__ counts, int max_per_line) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ctc_smem[];   // the line's frame labels and maxima: the run walk
    int32_t *flab = reinterpret_cast<int32_t *>(ctc_smem);                      // below is a chain of dependent reads
    float *fval = reinterpret_cast<float *>(ctc_smem) + T;
    const int n = blockIdx.x, lane = threadIdx.x;
    const int len = min(max(lens[n], 0), T);
    for (int t = lane; t < len; t += 64) {
        flab[t] = flab_all[(size_t)n * T + t];
        fval[t] = fval_all[(size_t)n * T + t];
    }
    __syncthreads();
    int emitted = 0;
    for (int base = 0; base < len; base += 64) {
        const int t = base + lane;
        const bool in = t < len;
        const int lab = in ? flab[t] : 0;
        const int prev = (in && t > 0) ? flab[t - 1] : -1;
        const bool open = in && lab != 0 && lab != prev;
        const unsigned long long mask = __ballot(open);
        const int slot = emitted + __popcll(mask & ((1ull << lane) - 1ull));
        if (open && slot < max_per_line) {
            int e = t;
            float mx = fval[t];
            while (e + 1 < len && flab[e + 1] == lab) { ++e; mx = fmaxf(mx, fval[e]); }
            const size_t o = (size_t)n * max_per_line + slot;
            labels[o] = lab; starts[o] = t; ends[o] = e; conf[o] = mx;
        }
        emitted += __popcll(mask);
    }
    if (lane == 0) counts[n] = emitted;
}
