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

// ---- prefix beam search ------------------------------------------------------------------------------------
// Semantics: oracle/ctc_ref.py::beam_decoder (kraken.lib.ctc_decoder.beam_decoder's algorithm on log-softmax(logits); the
// reference itself never calls a beam decoder).  One wave per line; per frame:
//   1. lp = log_softmax(logits[t]) (lanes stride the classes);
//   2. candidates: every live prefix i stays (p_b += total + lp[0]; p_nb += p_nb + lp[last]) or is extended by s >= 1
//      (p_nb = (s == last ? p_b : total) + lp[s]); an extension whose label sequence equals another live prefix q is folded
//      into q's stay candidate (prefix identity = 64-bit rolling hash of the labels); logaddexp = max + log1p(exp(-|d|));
//   3. the `beam` best candidates by logaddexp(p_b, p_nb) survive; ties keep creation order (position i*C + s; a folded
//      candidate takes the smaller of the two positions and the start frame of that creator);
//   4. back-pointers (parent, appended label) per frame and prefix let one lane rebuild the best prefix with the frames at
//      which its labels were appended; ends / confidences follow the oracle's post-pass.
#define COCR_BEAM_MAX 32

__device__ __forceinline__ float lse2(float a, float b) {
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    const float m = fmaxf(a, b);
    return m + log1pf(expf(-fabsf(a - b)));
}

__global__ __launch_bounds__(64) void ctc_beam_kernel(const float *__restrict__ logits, int T, int C, const int32_t *__restrict__ lens, int beam,
                                                      int32_t *__restrict__ labels, int32_t *__restrict__ starts, int32_t *__restrict__ ends,
                                                      float *__restrict__ conf, int32_t *__restrict__ counts, int max_per_line,
                                                      int32_t *__restrict__ bp_all, float *__restrict__ logz_all) {
    extern __shared__ __attribute__((aligned(16))) unsigned char beam_smem[];
    unsigned long long *hash = reinterpret_cast<unsigned long long *>(beam_smem);      // 8-byte aligned arrays first
    unsigned long long *nhash = hash + COCR_BEAM_MAX;
    float *pb = reinterpret_cast<float *>(nhash + COCR_BEAM_MAX), *pnb = pb + COCR_BEAM_MAX;   // live prefixes
    float *npb = pnb + COCR_BEAM_MAX, *npnb = npb + COCR_BEAM_MAX;   // next frame
    float *spb = npnb + COCR_BEAM_MAX, *spnb = spb + COCR_BEAM_MAX;  // stay candidates
    float *mval = spnb + COCR_BEAM_MAX;                              // folded extension score per live prefix
    int *mpos = reinterpret_cast<int *>(mval + COCR_BEAM_MAX);       // ... and its position
    int *last = mpos + COCR_BEAM_MAX, *nlast = last + COCR_BEAM_MAX;
    int *sel = nlast + COCR_BEAM_MAX;                                // selected positions of this frame
    float *lp = reinterpret_cast<float *>(sel + COCR_BEAM_MAX);      // [C]
    float *cand = lp + C;                                            // [beam][C] candidate scores (-inf = none / taken)

    const int n = blockIdx.x, lane = threadIdx.x;
    const int len = min(max(lens[n], 0), T);
    const float *lg = logits + (size_t)n * T * C;
    int32_t *bp = bp_all + (size_t)n * T * COCR_BEAM_MAX;            // (parent << 16) | label, per frame and rank
    float *logz = logz_all + (size_t)n * T;
    if (lane == 0) { pb[0] = 0.f; pnb[0] = -INFINITY; last[0] = 0; hash[0] = 1469598103934665603ull; }
    int nb = 1;
    for (int t = 0; t < len; ++t) {
        // 1. log-softmax of the frame
        float mx = -INFINITY;
        for (int c = lane; c < C; c += 64) mx = fmaxf(mx, lg[(size_t)t * C + c]);
        mx = wave_max(mx);
        float sm = 0.f;
        for (int c = lane; c < C; c += 64) sm += expf(lg[(size_t)t * C + c] - mx);
        const float lz = mx + logf(wave_sum(sm));
        for (int c = lane; c < C; c += 64) lp[c] = (lg[(size_t)t * C + c] - mx) - (lz - mx);
        if (lane == 0) logz[t] = lz;
        if (lane < COCR_BEAM_MAX) { mval[lane] = -INFINITY; mpos[lane] = 0x7fffffff; }
        __syncthreads();
        // 2. candidates
        for (int i = 0; i < nb; ++i) {
            const float p_b = pb[i], p_nb = pnb[i], tot = lse2(p_b, p_nb);
            const int li = last[i];
            const unsigned long long hi = hash[i];
            for (int c = lane; c < C; c += 64) {
                float e = -INFINITY;
                if (c == 0) {
                    spb[i] = tot + lp[0];
                    spnb[i] = li > 0 ? p_nb + lp[li] : -INFINITY;
                } else {
                    e = (c == li ? p_b : tot) + lp[c];
                    if (e > -INFINITY) {
                        const unsigned long long hc = hi * 1099511628211ull + (unsigned long long)(c + 1);
                        for (int q = 0; q < nb; ++q)
                            if (q != i && hash[q] == hc) { mval[q] = e; mpos[q] = i * C + c; e = -INFINITY; break; }
                    }
                }
                cand[(size_t)i * C + c] = e;
            }
        }
        __syncthreads();
        if (lane < nb) {      // stay candidates, with a folded extension if there is one
            const float s_nb = lse2(spnb[lane], mval[lane]);
            spnb[lane] = s_nb;
            cand[(size_t)lane * C] = lse2(spb[lane], s_nb);
        }
        __syncthreads();
        // 3. the `beam` best by (score desc, position asc); a stay candidate's position is min(i*C, folded position)
        int nsel = 0;
        for (int r = 0; r < beam; ++r) {
            float best = -INFINITY;
            int bpos = 0x7fffffff, bidx = -1;
            for (int i = 0; i < nb; ++i)
                for (int c = lane; c < C; c += 64) {
                    const float v = cand[(size_t)i * C + c];
                    const int pos = c == 0 ? min(i * C, mpos[i]) : i * C + c;
                    if (v > best || (v == best && v > -INFINITY && pos < bpos)) { best = v; bpos = pos; bidx = i * C + c; }
                }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(best, o, 64);
                const int op = __shfl_xor(bpos, o, 64), oi = __shfl_xor(bidx, o, 64);
                if (ov > best || (ov == best && op < bpos)) { best = ov; bpos = op; bidx = oi; }
            }
            if (best == -INFINITY) break;
            if (lane == 0) { sel[r] = bidx; cand[bidx] = -INFINITY; }
            ++nsel;
            __syncthreads();
        }
        // 4. next beam + back-pointers
        if (lane < nsel) {
            const int idx = sel[lane], i = idx / C, c = idx - i * C;
            if (c == 0) {
                npb[lane] = spb[i]; npnb[lane] = spnb[i]; nlast[lane] = last[i]; nhash[lane] = hash[i];
                const int mp = mpos[i];
                bp[(size_t)t * COCR_BEAM_MAX + lane] = mp < i * C ? (((mp / C) << 16) | (mp % C)) : (i << 16);   // first creator
            } else {
                npb[lane] = -INFINITY; npnb[lane] = (c == last[i] ? pb[i] : lse2(pb[i], pnb[i])) + lp[c]; nlast[lane] = c;
                nhash[lane] = hash[i] * 1099511628211ull + (unsigned long long)(c + 1);
                bp[(size_t)t * COCR_BEAM_MAX + lane] = (i << 16) | c;
            }
        }
        __syncthreads();
        if (lane < nsel) { pb[lane] = npb[lane]; pnb[lane] = npnb[lane]; last[lane] = nlast[lane]; hash[lane] = nhash[lane]; }
        nb = nsel;
        __syncthreads();
    }
    // ---- best prefix: walk the back-pointers (one lane), then ends / confidences in parallel over the labels
    __shared__ int s_cnt;
    int32_t *olab = labels + (size_t)n * max_per_line, *ost = starts + (size_t)n * max_per_line;
    if (lane == 0) {
        int cnt = 0, e = 0;
        for (int t = len - 1; t >= 0; --t) {
            const int v = bp[(size_t)t * COCR_BEAM_MAX + e];
            if (v & 0xffff) ++cnt;
            e = v >> 16;
        }
        s_cnt = cnt;
        int k = cnt;
        e = 0;
        for (int t = len - 1; t >= 0; --t) {
            const int v = bp[(size_t)t * COCR_BEAM_MAX + e];
            if (v & 0xffff) { --k; if (k < max_per_line) { olab[k] = v & 0xffff; ost[k] = t; } }
            e = v >> 16;
        }
        counts[n] = cnt;
    }
    __syncthreads();
    __threadfence_block();
    const int cnt = min(s_cnt, max_per_line);
    for (int k = lane; k < cnt; k += 64) {
        const int c = olab[k], s = ost[k], limit = k + 1 < cnt ? ost[k + 1] : len;
        int e = s;
        while (e + 1 < limit && lg[(size_t)(e + 1) * C + c] > lg[(size_t)(e + 1) * C]) ++e;
        float mxp = -INFINITY;
        for (int t = s; t <= e; ++t) mxp = fmaxf(mxp, lg[(size_t)t * C + c] - logz[t]);
        ends[(size_t)n * max_per_line + k] = e;
        conf[(size_t)n * max_per_line + k] = expf(mxp);
    }
}
