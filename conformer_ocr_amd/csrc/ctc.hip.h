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


// ---- prefix beam search, restructured (same semantics and tie rules as ctc_beam_kernel above) ---------------------------
// The kernel above evaluates all beam x C candidates of a frame and selects with `beam` full scans: 150 us per frame.  Two
// observations remove almost all of that work without changing any decision:
//   * log-softmax and the ranking of the classes of a frame do not depend on the beam: phase A computes lp[t][:] and the
//     K = min(beam + 1, C - 1) best non-blank classes of every frame (ties: smaller class first) with all four waves, frames
//     in parallel;
//   * an extension (i, c) scores tot_i + lp[c] (p_b,i + lp[c] if c repeats prefix i's last label), so within one parent the
//     extensions rank like the classes: an extension outside the K best classes has at least beam + 1 better-ranked extensions
//     of the SAME parent (same score offset or better, smaller position on ties) and can never be among the `beam` survivors.
//     Phase B (one wave, frames in order) therefore ranks beam x K + beam candidates instead of beam x C.  Folding (an extension
//     that equals another live prefix q adds to q's stay candidate, whatever its class rank) is found from q's side: the only
//     possible parent is the live prefix whose hash, extended by q's last label, equals q's hash.
// Keys: (monotone u32 image of the score) << 32 | ~position -- one 64-bit wave maximum per selection round, by DPP.
#define COCR_BEAM_KMAX (COCR_BEAM_MAX + 1)

__device__ __forceinline__ unsigned long long beam_key(float score, unsigned pos) {
    if (!(score > -INFINITY)) return 0ull;
    unsigned u = __builtin_bit_cast(unsigned, score);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;
    return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - pos);
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ unsigned long long dpp_max_u64(unsigned long long v) {
    const int lo = (int)(unsigned)v, hi = (int)(unsigned)(v >> 32);
    const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xF, false);
    const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xF, false);
    const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
    return o > v ? o : v;
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ unsigned dpp_max_u32(unsigned v) {
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xF, false);
    return o > v ? o : v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = dpp_max_u32<0xB1, 0xF>(v);
    v = dpp_max_u32<0x4E, 0xF>(v);
    v = dpp_max_u32<0x141, 0xF>(v);
    v = dpp_max_u32<0x140, 0xF>(v);
    v = dpp_max_u32<0x142, 0xA>(v);
    v = dpp_max_u32<0x143, 0xC>(v);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// maximum of 64-bit keys as two 32-bit maxima: the score word first, then the position word among the lanes that hold it
__device__ __forceinline__ unsigned long long wave_max_key(unsigned long long v) {
    const unsigned hi = (unsigned)(v >> 32), mhi = wave_max_u32(hi);
    const unsigned long long owners = __builtin_amdgcn_ballot_w64(hi == mhi);
    unsigned mlo;
    if (__builtin_popcountll(owners) == 1)                   // the usual case: one lane holds the best score -- its position by v_readlane
        mlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, (int)__builtin_ctzll(owners));
    else
        mlo = wave_max_u32(hi == mhi ? (unsigned)v : 0u);    // equal scores: the smallest position (largest ~position) wins
    return ((unsigned long long)mhi << 32) | mlo;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    v = dpp_max_u64<0xB1, 0xF>(v);
    v = dpp_max_u64<0x4E, 0xF>(v);
    v = dpp_max_u64<0x141, 0xF>(v);
    v = dpp_max_u64<0x140, 0xF>(v);
    v = dpp_max_u64<0x142, 0xA>(v);
    v = dpp_max_u64<0x143, 0xC>(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}

// C <= 256.  Scratch per line: lp_all [T][C] f32, topc [T][COCR_BEAM_KMAX] i32, bp [T][COCR_BEAM_MAX] i32, logz [T] f32.
__global__ __launch_bounds__(256) void ctc_beam2_kernel(const float *__restrict__ logits, int T, int C, const int32_t *__restrict__ lens, int beam,
                                                        int32_t *__restrict__ labels, int32_t *__restrict__ starts, int32_t *__restrict__ ends,
                                                        float *__restrict__ conf, int32_t *__restrict__ counts, int max_per_line,
                                                        float *__restrict__ lp_all_, int32_t *__restrict__ topc_all, int32_t *__restrict__ bp_all,
                                                        float *__restrict__ logz_all, unsigned long long *dbg) {
    constexpr int NB = COCR_BEAM_MAX, KM = COCR_BEAM_KMAX;
    __shared__ unsigned long long hash[NB], nhash[NB], ckey[NB * KM];
    __shared__ float pb[NB], pnb[NB], tot[NB], spb[NB], spnb[NB], mval[NB], lpl[NB], npb[NB], npnb[NB], s_tl[KM];
    __shared__ int last[NB], nlast[NB], mpos[NB], sel[NB], s_tc[KM];
    __shared__ int s_cnt;
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int len = min(max(lens[n], 0), T);
    const float *lg = logits + (size_t)n * T * C;
    float *lp_all = lp_all_ + (size_t)n * T * C;
    int32_t *topc = topc_all + (size_t)n * T * KM;
    int32_t *bp = bp_all + (size_t)n * T * NB;
    float *logz = logz_all + (size_t)n * T;
    const int K = min(beam + 1, C - 1);

    // ---- phase A: frames in parallel (4 waves): log-softmax and the K best non-blank classes of each frame
    for (int t = wave; t < len; t += 4) {
        float x[4], mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int c = lane + 64 * j; x[j] = c < C ? lg[(size_t)t * C + c] : -INFINITY; mx = fmaxf(mx, x[j]); }
        mx = wave_max(mx);
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (lane + 64 * j < C) sm += expf(x[j] - mx);
        const float lz = mx + logf(wave_sum(sm));
        unsigned long long key[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = lane + 64 * j;
            const float l = (x[j] - mx) - (lz - mx);
            if (c < C) lp_all[(size_t)t * C + c] = l;
            key[j] = (c >= 1 && c < C) ? beam_key(l, (unsigned)c) : 0ull;      // (-inf lp: key 0, never ranked)
        }
        if (lane == 0) logz[t] = lz;
        for (int r = 0; r < K; ++r) {
            unsigned long long best = key[0];
#pragma unroll
            for (int j = 1; j < 4; ++j) best = key[j] > best ? key[j] : best;
            const unsigned long long top = wave_max_key(best);
#pragma unroll
            for (int j = 0; j < 4; ++j) if (key[j] == top && top) key[j] = 0ull;
            if (lane == 0) topc[(size_t)t * KM + r] = top ? (int)(0xFFFFFFFFu - (unsigned)top) : 0;      // 0 = no further class with finite score
        }
    }
    __threadfence_block();
    __syncthreads();
    if (wave != 0) return;

    // ---- phase B: one wave walks the frames
    __shared__ unsigned long long hx[NB];                    // hash[i] * prime: an extension's hash is hx[i] + (class + 1)
    if (lane == 0) { pb[0] = 0.f; pnb[0] = -INFINITY; last[0] = 0; hash[0] = 1469598103934665603ull; }
    int nb = 1;
    auto lds_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); };
    // this lane's extension candidates j = lane + 64 s: (parent j / K, class rank j % K); in registers for up to SLOTS per lane
    constexpr int SLOTS = 6;
    const bool in_regs = beam * K <= 64 * SLOTS;
    int ci[SLOTS], cr[SLOTS];
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) { const int j = lane + 64 * sl; ci[sl] = j / K; cr[sl] = j - ci[sl] * K; }
    // The frame's log-softmax row and class ranking live in LDS, one frame ahead (global latency would otherwise sit on the
    // serial path twice per frame: ranking -> lp[class])
    __shared__ float lp_s[2][256];
    __shared__ int tc_s[2][KM];
    float nlp[4];
    int ntc = 0;
    auto request_frame = [&](int t) {
        const int tt = min(t, max(len - 1, 0));
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int c = lane + 64 * j; nlp[j] = c < C ? lp_all[(size_t)tt * C + c] : 0.f; }
        ntc = lane < K ? topc[(size_t)tt * KM + lane] : 0;
    };
    auto publish_frame = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) lp_s[buf][lane + 64 * j] = nlp[j];
        if (lane < KM) tc_s[buf][lane] = ntc;
    };
    request_frame(0);
    publish_frame(0);
    lds_sync();
#ifdef COCR_CHAIN_STAMPS_BUILD
    unsigned long long acc_t[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#define BSTAMP(k) { const unsigned long long now = __builtin_readcyclecounter(); acc_t[k] += now - tprev; tprev = now; }
#else
#define BSTAMP(k)
#endif
    for (int t = 0; t < len; ++t) {
        const float *lp = lp_s[t & 1];
        request_frame(t + 1);                                    // in flight during this frame
        const float lp0 = lp[0];
        if (lane < K) { const int c = tc_s[t & 1][lane]; s_tc[lane] = c; s_tl[lane] = c ? lp[c] : -INFINITY; }
        if (lane < nb) {
            const float p_b = pb[lane], p_nb = pnb[lane], tt = lse2(p_b, p_nb);
            const int li = last[lane];
            const float l = li > 0 ? lp[li] : 0.f;
            tot[lane] = tt; lpl[lane] = l;
            spb[lane] = tt + lp0;
            spnb[lane] = li > 0 ? p_nb + l : -INFINITY;
            mval[lane] = -INFINITY; mpos[lane] = 0x7fffffff;
            hx[lane] = hash[lane] * 1099511628211ull;
        }
        lds_sync();
        BSTAMP(0)
        // folds: extension (i, last_q) of parent i equals live prefix q
        for (int pi = lane; pi < beam * beam; pi += 64) {
            const int q = pi / beam, i = pi - q * beam;              // (the compiler hoists this out of the frame loop)
            if (q >= nb || i >= nb) continue;
            const int lq = last[q];
            if (i != q && lq > 0 && hx[i] + (unsigned long long)(lq + 1) == hash[q]) {
                const float add = (lq == last[i] ? pb[i] : tot[i]) + lpl[q];
                if (add > -INFINITY) { mval[q] = add; mpos[q] = i * C + lq; }
            }
        }
        lds_sync();
        BSTAMP(1)
        // candidates: extensions (i, r-th class) unless folded; then this lane's stay candidate
        const int ncand = nb * K;
        unsigned long long ck[SLOTS];
        auto ext_key = [&](int i, int r) -> unsigned long long {
            const int c = s_tc[r];
            if (c <= 0) return 0ull;
            float e = (c == last[i] ? pb[i] : tot[i]) + s_tl[r];
            const int pos = i * C + c;
            for (int q = 0; q < nb; ++q) if (mpos[q] == pos) e = -INFINITY;             // folded into q's stay candidate
            return beam_key(e, (unsigned)pos);
        };
        if (in_regs) {
            float ce[SLOTS];
            int cpos[SLOTS];
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
                const int i = ci[sl], c = lane + 64 * sl < ncand ? s_tc[cr[sl]] : 0;
                cpos[sl] = i * C + c;
                ce[sl] = c > 0 ? (c == last[i] ? pb[i] : tot[i]) + s_tl[cr[sl]] : -INFINITY;
            }
            const int my_mpos = lane < nb ? mpos[lane] : 0x7fffffff;                   // folded positions: lane q -> everyone, by v_readlane
            for (int q = 0; q < nb; ++q) {
                const int mp = __builtin_amdgcn_readlane(my_mpos, q);
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl) if (cpos[sl] == mp) ce[sl] = -INFINITY;
            }
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) ck[sl] = beam_key(ce[sl], (unsigned)cpos[sl]);
            // sort the lane's keys (descending): a selection round then looks at ck[0] only
#pragma unroll
            for (int a2 = 0; a2 < SLOTS; ++a2)
#pragma unroll
                for (int b2 = 0; b2 + 1 < SLOTS - a2; ++b2)
                    if (ck[b2] < ck[b2 + 1]) { const unsigned long long tmp = ck[b2]; ck[b2] = ck[b2 + 1]; ck[b2 + 1] = tmp; }
        } else {
            for (int j = lane; j < ncand; j += 64) { const int i = j / K; ckey[j] = ext_key(i, j - i * K); }
        }
        unsigned long long skey = 0ull;
        if (lane < nb) {
            const float s_nb = lse2(spnb[lane], mval[lane]);
            spnb[lane] = s_nb;
            skey = beam_key(lse2(spb[lane], s_nb), (unsigned)min(lane * C, mpos[lane]));
        }
        lds_sync();
        BSTAMP(2)
        // the `beam` best keys
        int nsel = 0;
        if (in_regs) {
            // Branch-free rounds: the lane's keys (extensions + its stay candidate) are one sorted list; a round is two 32-bit DPP
            // maxima over the heads, the owner pops by selects, lane r keeps the r-th key.  Who the key belongs to is decoded
            // afterwards from its position: class 0 = stay of prefix pos / C; a folded position = stay of the prefix that absorbed
            // it; anything else = extension (pos / C, pos % C).
            unsigned long long lk[SLOTS + 1];
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) lk[sl] = ck[sl];
            lk[SLOTS] = skey;
#pragma unroll
            for (int sl = SLOTS; sl > 0; --sl)                    // insert the stay key into the sorted extensions
                if (lk[sl] > lk[sl - 1]) { const unsigned long long tmp = lk[sl]; lk[sl] = lk[sl - 1]; lk[sl - 1] = tmp; }
            unsigned long long mykey = 0ull;
            for (int r = 0; r < beam; ++r) {
                const unsigned long long top = wave_max_key(lk[0]);
                if (!top) break;                                  // uniform
                const bool mine = lk[0] == top;
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl) lk[sl] = mine ? lk[sl + 1] : lk[sl];
                lk[SLOTS] = mine ? 0ull : lk[SLOTS];
                mykey = lane == r ? top : mykey;
                ++nsel;
            }
            const int pos = (int)(0xFFFFFFFFu - (unsigned)mykey), pi2 = pos / C, pc = pos - pi2 * C;
            int idx = pc == 0 ? 0x40000000 + pi2 : pos;
            const int my_mpos2 = lane < nb ? mpos[lane] : 0x7fffffff;
            for (int q = 0; q < nb; ++q) if (__builtin_amdgcn_readlane(my_mpos2, q) == pos) idx = 0x40000000 + q;
            if (lane < nsel) sel[lane] = idx;
        } else {
            for (int r = 0; r < beam; ++r) {
                unsigned long long best = skey;
                int bidx = 0x40000000 + lane;
                for (int j = lane; j < ncand; j += 64) { const unsigned long long k2 = ckey[j]; if (k2 > best) { best = k2; bidx = j; } }
                const unsigned long long top = wave_max_key(best);
                if (!top) break;
                if (best == top) {                               // keys are unique (positions are): exactly one lane
                    sel[r] = bidx;
                    if (bidx >= 0x40000000) skey = 0ull; else ckey[bidx] = 0ull;
                }
                ++nsel;
                lds_sync();
            }
        }
        lds_sync();
        BSTAMP(3)
        // next beam + back-pointers
        if (lane < nsel) {
            const int idx = sel[lane];
            if (idx >= 0x40000000) {
                const int q = idx - 0x40000000, mp = mpos[q];
                npb[lane] = spb[q]; npnb[lane] = spnb[q]; nlast[lane] = last[q]; nhash[lane] = hash[q];
                bp[(size_t)t * NB + lane] = mp < q * C ? (((mp / C) << 16) | (mp % C)) : (q << 16);      // first creator
            } else {
                int i, c;
                if (in_regs) { i = idx / C; c = idx - i * C; }             // a position
                else { i = idx / K; c = s_tc[idx - i * K]; }               // a candidate slot
                npb[lane] = -INFINITY; npnb[lane] = (c == last[i] ? pb[i] : tot[i]) + lp[c]; nlast[lane] = c;
                nhash[lane] = hx[i] + (unsigned long long)(c + 1);
                bp[(size_t)t * NB + lane] = (i << 16) | c;
            }
        }
        lds_sync();
        if (lane < nsel) { pb[lane] = npb[lane]; pnb[lane] = npnb[lane]; last[lane] = nlast[lane]; hash[lane] = nhash[lane]; }
        nb = nsel;
        publish_frame((t + 1) & 1);
        lds_sync();
        BSTAMP(4)
    }
#ifdef COCR_CHAIN_STAMPS_BUILD
    if (dbg && n == 0 && lane == 0) for (int k2 = 0; k2 < 5; ++k2) dbg[k2] = acc_t[k2] + 1;
#endif
    // ---- best prefix: walk the back-pointers (one lane), then ends / confidences in parallel over the labels
    int32_t *olab = labels + (size_t)n * max_per_line, *ost = starts + (size_t)n * max_per_line;
    if (lane == 0) {
        int cnt = 0, e = 0;
        for (int t = len - 1; t >= 0; --t) {
            const int v = bp[(size_t)t * NB + e];
            if (v & 0xffff) ++cnt;
            e = v >> 16;
        }
        s_cnt = cnt;
        int k = cnt;
        e = 0;
        for (int t = len - 1; t >= 0; --t) {
            const int v = bp[(size_t)t * NB + e];
            if (v & 0xffff) { --k; if (k < max_per_line) { olab[k] = v & 0xffff; ost[k] = t; } }
            e = v >> 16;
        }
        counts[n] = cnt;
    }
    __threadfence_block();
    lds_sync();
    const int cnt = min(s_cnt, max_per_line);
    for (int k = lane; k < cnt; k += 64) {
        const int c = olab[k], s0 = ost[k], limit = k + 1 < cnt ? ost[k + 1] : len;
        int e = s0;
        while (e + 1 < limit && lg[(size_t)(e + 1) * C + c] > lg[(size_t)(e + 1) * C]) ++e;
        float mxp = -INFINITY;
        for (int t = s0; t <= e; ++t) mxp = fmaxf(mxp, lg[(size_t)t * C + c] - logz[t]);
        ends[(size_t)n * max_per_line + k] = e;
        conf[(size_t)n * max_per_line + k] = expf(mxp);
    }
}
