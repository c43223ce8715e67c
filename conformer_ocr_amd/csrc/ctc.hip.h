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

// logaddexp on the hardware's exp2 / log2 (~1 ulp each): max + ln 2 * log2(1 + 2^(-|a - b| log2 e)).  The beam kernels sit on a
// serial chain of these (two per frame and prefix); the libm log1pf(expf()) form is ~50 dependent instructions, this one 8.
// -inf operands: |a - b| = inf gives max + 0; a == b (also both -inf) gives max + ln 2 without forming inf - inf.
__device__ __forceinline__ float lse2(float a, float b) {
    const float m = fmaxf(a, b), d = a == b ? 0.f : -fabsf(a - b);
    return m + 0.6931471805599453f * __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(d * 1.4426950408889634f));
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
// The kernel above evaluates all beam x C candidates of a frame and selects with `beam` full scans: 150 us per frame.  Three
// observations remove almost all of that work without changing any decision:
//   * log-softmax and the ranking of the classes of a frame do not depend on the beam: ctc_beam_rank_kernel (one workgroup per
//     frame, the whole batch in parallel) writes a 2 KB record per frame: lp[256], the rank of every class, the
//     K = min(beam + 1, C - 1) best non-blank classes (ties: smaller class first) with their lp, and log Z;
//   * an extension (i, c) scores tot_i + lp[c] (p_b,i + lp[c] if c repeats prefix i's last label) and the live prefixes are
//     sorted by tot (they are the previous frame's survivors in rank order).  Extension (i, r) -- prefix i, the class of rank r
//     -- is therefore beaten by every (i', r') with i' <= i, r' <= r other than itself (score >= in float arithmetic, which is
//     monotone, and a smaller position i' * C + c' on ties), except by the at most one repeat-demoted extension per prefix; an
//     extension that was folded away is replaced in that count by the stay candidate that absorbed it (score >=, position <=).
//     At least (i + 1) * r - 1 candidates rank above it, so it can survive only if (i + 1) * r <= beam: for beam 16 that is a
//     STATIC set of 66 extensions; with the 16 stay candidates, 82 candidates per frame instead of 16 x 256;
//   * the survivors are found without any serial selection round: every lane holds its candidates' 64-bit keys
//     ((monotone image of the score) << 32 | ~position: unique), all keys go through LDS once, and a candidate's rank is the
//     number of larger keys (one v_cmp_gt_u64 + add per pair).  The candidate of rank R < beam writes the state of next frame's
//     prefix R itself (including what that prefix needs from the next frame's record), so a frame is three LDS round trips.
// Folding (an extension that equals another live prefix q adds to q's stay candidate, whatever its class rank) is found from
// q's side: the only possible parent is the live prefix whose hash equals the hash of q without its last label; q's lane then
// clears that extension's key in LDS (its slot follows from the rank of q's last label in the frame's record).
// The walk kernel is one wave per line; frame records stream into a 4-deep LDS ring by LDS-DMA three frames ahead, and the
// back-pointers stay in LDS (global memory when T * beam is too large for that), so no global latency is on the serial path.
#define COCR_BEAM_KMAX (COCR_BEAM_MAX + 1)
#define COCR_BEAM_REC 2048          // bytes per (line, frame): lp[256] f32 | rank[256] u8 | top class[64] i32 | top lp[64] f32 | log Z, pad
#define COCR_BEAM_REC_RK 1024
#define COCR_BEAM_REC_TC 1280
#define COCR_BEAM_REC_TL 1536
#define COCR_BEAM_REC_LZ 1792

__device__ __forceinline__ unsigned long long beam_key(float score, unsigned pos) {
    if (!(score > -INFINITY)) return 0ull;
    unsigned u = __builtin_bit_cast(unsigned, score);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;
    return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - pos);
}

// C <= 256.  One workgroup per frame of the batch; thread c owns class c.
__global__ __launch_bounds__(256) void ctc_beam_rank_kernel(const float *__restrict__ logits, int T, int C, const int32_t *__restrict__ lens, int K,
                                                            unsigned char *__restrict__ rec_all) {
    __shared__ __attribute__((aligned(16))) unsigned long long key_s[256];
    __shared__ float e_s[256], tl_s[64];
    __shared__ int tc_s[64];
    const int f = blockIdx.x, n = f / T, t = f - n * T, c = threadIdx.x, lane = c & 63;
    if (t >= min(max(lens[n], 0), T)) return;
    unsigned char *rec = rec_all + (size_t)f * COCR_BEAM_REC;
    const float x = c < C ? logits[(size_t)f * C + c] : -INFINITY;
    e_s[c] = x;
    if (c < 64) { tc_s[c] = 0; tl_s[c] = -INFINITY; }
    __syncthreads();
    // the frame's maximum and sum in the lane-strided order of ctc_beam_kernel (every wave computes both: no second exchange)
    float xs[4], mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) { xs[j] = e_s[lane + 64 * j]; mx = fmaxf(mx, xs[j]); }
    mx = wave_max(mx);
    float sm = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) if (lane + 64 * j < C) sm += expf(xs[j] - mx);
    const float lz = mx + logf(wave_sum(sm));
    const float l = c < C ? (x - mx) - (lz - mx) : -INFINITY;
    reinterpret_cast<float *>(rec)[c] = l;
    const unsigned long long key = (c >= 1 && c < C) ? beam_key(l, (unsigned)c) : 0ull;      // (-inf lp: key 0, never ranked)
    key_s[c] = key;
    __syncthreads();
    int rank = 0;
#pragma unroll 8
    for (int j = 0; j < 256; j += 2) {
        const ulonglong2 kk = *reinterpret_cast<const ulonglong2 *>(&key_s[j]);
        rank += (kk.x > key) + (kk.y > key);
    }
    rec[COCR_BEAM_REC_RK + c] = (unsigned char)(key ? min(rank, 255) : 255);
    if (key && rank < K) { tc_s[rank] = c; tl_s[rank] = l; }
    __syncthreads();
    if (c < 64) {
        reinterpret_cast<int32_t *>(rec + COCR_BEAM_REC_TC)[c] = tc_s[c];                    // 0 = no further class with a finite score
        reinterpret_cast<float *>(rec + COCR_BEAM_REC_TL)[c] = tl_s[c];
    }
    if (c == 0) *reinterpret_cast<float *>(rec + COCR_BEAM_REC_LZ) = lz;
}

// One workgroup of four waves per line, one candidate per thread: threads [0, NE) own the extensions (waves 0..2; NE <= 151 for
// beam 32), threads 192 + q the stay candidate of prefix q (wave 3).  Three barriers per frame: (A) next frame's prefixes are in
// LDS -> all 256 threads look for fold pairs (one (q, i) hash comparison each for beam <= 16) -> (A2) -> keys -> (B) -> ranks ->
// the survivors write the prefixes.  Dynamic LDS: the line's back-pointers [T][beam] and the label stack of the final walk
// [T][2] when bp_in_lds, else nothing (bp_gbl: [N][T][COCR_BEAM_MAX]).
// One wave alone on its SIMD pays ~7 cycles per DEPENDENT vector instruction and ~8 per compare into a scalar register pair
// (tools/micro/sgpr_dep.hip), so the rank loop is written out: per step eight independent 64-bit compares into eight scalar
// pairs, eight add-with-carry, and the next eight keys' LDS reads in flight underneath (15 cycles per key; the compiler's
// compare / select / add chain: 21, and it waits for every read).
typedef unsigned long long beam_u64x2 __attribute__((ext_vector_type(2)));
template <bool LOAD>
__device__ __forceinline__ void beam_rank_step(int &rank, int &rank2, beam_u64x2 (&nx)[4], const beam_u64x2 (&cu)[4], unsigned long long key, unsigned next_addr) {
    if (LOAD)
        asm volatile("ds_read_b128 %2, %14\n\tds_read_b128 %3, %14 offset:16\n\tds_read_b128 %4, %14 offset:32\n\tds_read_b128 %5, %14 offset:48\n\t"
                     "v_cmp_gt_u64 s[40:41], %6, %15\n\tv_cmp_gt_u64 s[42:43], %7, %15\n\tv_cmp_gt_u64 s[44:45], %8, %15\n\tv_cmp_gt_u64 s[46:47], %9, %15\n\t"
                     "v_cmp_gt_u64 s[48:49], %10, %15\n\tv_cmp_gt_u64 s[50:51], %11, %15\n\tv_cmp_gt_u64 s[52:53], %12, %15\n\tv_cmp_gt_u64 s[54:55], %13, %15\n\t"
                     "v_addc_co_u32 %0, vcc, 0, %0, s[40:41]\n\tv_addc_co_u32 %1, vcc, 0, %1, s[42:43]\n\tv_addc_co_u32 %0, vcc, 0, %0, s[44:45]\n\t"
                     "v_addc_co_u32 %1, vcc, 0, %1, s[46:47]\n\tv_addc_co_u32 %0, vcc, 0, %0, s[48:49]\n\tv_addc_co_u32 %1, vcc, 0, %1, s[50:51]\n\t"
                     "v_addc_co_u32 %0, vcc, 0, %0, s[52:53]\n\tv_addc_co_u32 %1, vcc, 0, %1, s[54:55]\n\ts_waitcnt lgkmcnt(0)"
                     : "+v"(rank), "+v"(rank2), "=&v"(nx[0]), "=&v"(nx[1]), "=&v"(nx[2]), "=&v"(nx[3])
                     : "v"(cu[0].x), "v"(cu[0].y), "v"(cu[1].x), "v"(cu[1].y), "v"(cu[2].x), "v"(cu[2].y), "v"(cu[3].x), "v"(cu[3].y), "v"(next_addr), "v"(key)
                     : "vcc", "memory", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55");
    else
        asm volatile("v_cmp_gt_u64 s[40:41], %2, %10\n\tv_cmp_gt_u64 s[42:43], %3, %10\n\tv_cmp_gt_u64 s[44:45], %4, %10\n\tv_cmp_gt_u64 s[46:47], %5, %10\n\t"
                     "v_cmp_gt_u64 s[48:49], %6, %10\n\tv_cmp_gt_u64 s[50:51], %7, %10\n\tv_cmp_gt_u64 s[52:53], %8, %10\n\tv_cmp_gt_u64 s[54:55], %9, %10\n\t"
                     "v_addc_co_u32 %0, vcc, 0, %0, s[40:41]\n\tv_addc_co_u32 %1, vcc, 0, %1, s[42:43]\n\tv_addc_co_u32 %0, vcc, 0, %0, s[44:45]\n\t"
                     "v_addc_co_u32 %1, vcc, 0, %1, s[46:47]\n\tv_addc_co_u32 %0, vcc, 0, %0, s[48:49]\n\tv_addc_co_u32 %1, vcc, 0, %1, s[50:51]\n\t"
                     "v_addc_co_u32 %0, vcc, 0, %0, s[52:53]\n\tv_addc_co_u32 %1, vcc, 0, %1, s[54:55]"
                     : "+v"(rank), "+v"(rank2)
                     : "v"(cu[0].x), "v"(cu[0].y), "v"(cu[1].x), "v"(cu[1].y), "v"(cu[2].x), "v"(cu[2].y), "v"(cu[3].x), "v"(cu[3].y), "v"(key)
                     : "vcc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55");
}
__device__ __forceinline__ bool beam_eq64(unsigned long long a, unsigned long long b) {
    return (((unsigned)a ^ (unsigned)b) | ((unsigned)(a >> 32) ^ (unsigned)(b >> 32))) == 0u;
}

template <int SL>
__global__ __launch_bounds__(256) void ctc_beam_walk_kernel(const float *__restrict__ logits, int T, int C, const int32_t *__restrict__ lens, int beam, int K,
                                                            int32_t *__restrict__ labels, int32_t *__restrict__ starts, int32_t *__restrict__ ends,
                                                            float *__restrict__ conf, int32_t *__restrict__ counts, int max_per_line,
                                                            const unsigned char *__restrict__ rec_all, int32_t *__restrict__ bp_gbl, int bp_in_lds,
                                                            unsigned long long *dbg) {
    constexpr int NB = COCR_BEAM_MAX, KM = COCR_BEAM_KMAX, RING = 4, STAY0 = 192;
    constexpr unsigned long long PRIME = 1099511628211ull;
    __shared__ __attribute__((aligned(16))) unsigned char ring[RING][COCR_BEAM_REC];
    __shared__ __attribute__((aligned(16))) unsigned long long ckey[STAY0 + NB];                 // extensions [0, nep), stays [nep, nep + beam), zeros between
    __shared__ __attribute__((aligned(16))) unsigned long long hash[NB], phash[NB], hx[NB];
    __shared__ __attribute__((aligned(16))) float pb[NB], tot[NB], spb[NB], spnb[NB], lpl[NB];
    __shared__ __attribute__((aligned(16))) int last[NB], pair_s[NB], live_s[4], off_s[KM + 1], rank_s[64 * SL], s_cnt;
    extern __shared__ __attribute__((aligned(16))) unsigned char beam_dyn[];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int len = min(max(lens[n], 0), T);
    const float *lg = logits + (size_t)n * T * C;
    const unsigned char *rec = rec_all + (size_t)n * T * COCR_BEAM_REC;
    int32_t *bpl = reinterpret_cast<int32_t *>(beam_dyn);                      // [T][beam]
    int32_t *bpg = bp_gbl + (size_t)n * T * NB;                                // [T][NB]

    // frame records: global -> LDS ring, two 1 KB LDS-DMA instructions per frame, issued three frames ahead by wave 2
    auto request = [&](int t) {
        const unsigned char *src = rec + (size_t)min(t, max(len - 1, 0)) * COCR_BEAM_REC + lane * 16;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(&ring[t & (RING - 1)][0]), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + 1024), (lds_ptr_t)(&ring[t & (RING - 1)][1024]), 16, 0, 0);
    };
    if (wave == 2 && len > 0) { request(0); request(1); request(2); }

    // the static candidate set: extension (i, r) is thread off[r] + i for i < cnt(r) = (r ? min(beam, beam / r) : beam)
    if (tid == 0) {
        int o = 0;
        for (int r = 0; r < K; ++r) { off_s[r] = o; o += r ? min(beam, beam / r) : beam; }
        off_s[K] = o;
    }
    if (tid < STAY0 + NB) ckey[tid] = 0ull;
    if (tid < NB) {
        const bool root = tid == 0;
        pb[tid] = root ? 0.f : -INFINITY; tot[tid] = root ? 0.f : -INFINITY; spb[tid] = -INFINITY; spnb[tid] = -INFINITY; lpl[tid] = 0.f;
        last[tid] = 0; hash[tid] = root ? 1469598103934665603ull : 0ull; phash[tid] = 0ull; hx[tid] = root ? 1469598103934665603ull * PRIME : 0ull;
        pair_s[tid] = -1;
    }
    if (tid < 4) live_s[tid] = 0;
    if (tid < 64 * SL) rank_s[tid] = 0;
    __syncthreads();
    const int NE = off_s[K], nep = (NE + 7) & ~7, nblk = (nep + ((beam + 7) & ~7)) >> 3;      // (the key array is zero beyond the candidates)
    int ci = -1, cr = -1, slot = 0;                                            // this thread's candidate: extension (ci, cr), or stay of prefix ci (cr = -1), or none
    if (tid < NE) { int r = 0; while (off_s[r + 1] <= tid) ++r; cr = r; ci = tid - off_s[r]; slot = tid; }
    else if (tid >= STAY0 && tid - STAY0 < beam) { ci = tid - STAY0; slot = nep + ci; }
    // fold pairs: thread (q, g) compares phash[q] with hash[g + G u]: 16 lanes per q and one comparison for beam <= 16, else 8 lanes and four
    const int G = beam <= 16 ? 16 : 8, fq = tid / G, fg = tid - fq * G, fn = beam <= 16 ? 1 : 4;
    const unsigned ckey_addr = (unsigned)(unsigned long long)(lds_ptr_t)(&ckey[0]);
    if (wave == 2 && len > 0) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                        // frame 0 has landed
        if (lane == 0) spb[0] = reinterpret_cast<const float *>(&ring[0][0])[0];                 // tot (0) + lp[blank] of frame 0
    }
    __syncthreads();
#ifdef COCR_CHAIN_STAMPS_BUILD
    unsigned long long acc_t[5] = {0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#define BSTAMP(k) { const unsigned long long now = __builtin_readcyclecounter(); acc_t[k] += now - tprev; tprev = now; }
#else
#define BSTAMP(k)
#endif

    for (int t = 0; t < len; ++t) {
        if (wave == 2) {                                                       // (the ring slot of frame t - 1 was last read before barrier B of that frame)
            request(t + 3);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                    // frames <= t + 1 have landed (in-order completion); the barriers publish that
        }
        const unsigned char *cur = ring[t & (RING - 1)], *nxt = ring[(t + 1) & (RING - 1)];
        const float *lpn = reinterpret_cast<const float *>(nxt);
        const int32_t *tcc = reinterpret_cast<const int32_t *>(cur + COCR_BEAM_REC_TC);
        const float *tlc = reinterpret_cast<const float *>(cur + COCR_BEAM_REC_TL);

        // ---- fold pairs: prefix q equals prefix i extended by q's last label (then that extension adds to q's stay candidate)
        {
            const int lq = last[fq];
            const unsigned long long ph = phash[fq];
            int iq = -1;
            for (int u = 0; u < fn; ++u) {
                const int i = fg + G * u;
                const bool hit = lq > 0 && i != fq && beam_eq64(hash[i], ph);
                const unsigned grp = (unsigned)(__builtin_amdgcn_ballot_w64(hit) >> (lane & ~(G - 1))) & ((1u << G) - 1u);
                if (grp) iq = G * u + (31 - __builtin_clz(grp));
            }
            if (fg == 0) pair_s[fq] = iq >= 0 ? ((iq << 16) | lq) : -1;
        }
        // ---- the candidate: everything that does not depend on the pairs; what a survivor will need stays in registers
        float n_pb = -INFINITY, n_pnb = -INFINITY, n_tot = -INFINITY, s_lpl = 0.f;
        int n_last = 0, n_bp = 0;
        unsigned long long n_hash = 0ull, n_ph = 0ull, key = 0ull;
        if (cr >= 0) {                                                         // extension of prefix i by the class of rank r
            const int i = ci, c = tcc[cr], li = last[i];
            n_tot = c > 0 ? (c == li ? pb[i] : tot[i]) + tlc[cr] : -INFINITY;
            n_hash = hx[i] + (unsigned long long)(c + 1); n_ph = hash[i];
            n_last = c; n_bp = (i << 16) | c;
        } else if (ci >= 0) {                                                  // prefix i stays
            const int i = ci;
            n_pb = spb[i]; n_pnb = spnb[i]; n_last = last[i]; n_hash = hash[i]; n_ph = phash[i]; s_lpl = lpl[i];
        }
        BSTAMP(0)
        __syncthreads();                                                       // (A2)
        if (cr >= 0) {
            bool folded = false;                                               // it equals a live prefix: not a candidate of its own
            int4 pr[SL == 2 ? 4 : 8];                                          // (SL == 2: beam <= 16; entries beyond the beam are -1; all reads first)
#pragma unroll
            for (int u = 0; u < (SL == 2 ? 4 : 8); ++u) pr[u] = *reinterpret_cast<const int4 *>(&pair_s[4 * u]);
#pragma unroll
            for (int u = 0; u < (SL == 2 ? 4 : 8); ++u) folded |= (pr[u].x == n_bp) | (pr[u].y == n_bp) | (pr[u].z == n_bp) | (pr[u].w == n_bp);
            if (folded) n_tot = -INFINITY;
            key = beam_key(n_tot, (unsigned)(ci * C + n_last));
            n_pnb = n_tot;
        } else if (ci >= 0) {
            const int i = ci, pr = pair_s[i];
            float mval = -INFINITY;
            int mpos = 0x7fffffff;
            if (pr >= 0) {
                const int iq = pr >> 16;
                const float add = (n_last == last[iq] ? pb[iq] : tot[iq]) + s_lpl;
                if (add > -INFINITY) { mval = add; mpos = iq * C + n_last; }
            }
            n_pnb = lse2(n_pnb, mval);
            n_tot = lse2(n_pb, n_pnb);
            key = beam_key(n_tot, (unsigned)min(i * C, mpos));
            n_bp = mpos < i * C ? pr : (i << 16);                              // first creator
        }
        if (ci >= 0) ckey[slot] = key;
        {
            const int cnt = __builtin_popcountll(__builtin_amdgcn_ballot_w64(key != 0ull));
            if (lane == 0) live_s[wave] = cnt;
        }
        BSTAMP(1)
        __syncthreads();                                                       // (B)
        // ---- rank = the number of larger keys.  Every wave takes a quarter of the key blocks (eight keys each) for ALL candidates
        // (lane l: candidates l, l + 64, ...): all 256 threads reading all keys would be 180 KB per frame through the LDS port.
        {
            unsigned long long k[SL];
            int rk[SL], rk2[SL];
#pragma unroll
            for (int sl = 0; sl < SL; ++sl) { k[sl] = ckey[lane + 64 * sl]; rk[sl] = 0; rk2[sl] = 0; }
            beam_u64x2 ka[4], kb[4];
            int b = wave;
            if (b < nblk) {
#pragma unroll
                for (int u = 0; u < 4; ++u) ka[u] = *reinterpret_cast<const beam_u64x2 *>(&ckey[8 * b + 2 * u]);
            }
            for (; b + 4 < nblk; b += 8) {                                     // (two steps per trip: the key registers alternate)
                beam_rank_step<true>(rk[0], rk2[0], kb, ka, k[0], ckey_addr + (b + 4) * 64);
#pragma unroll
                for (int sl = 1; sl < SL; ++sl) beam_rank_step<false>(rk[sl], rk2[sl], kb, ka, k[sl], 0u);
                if (b + 8 < nblk) {
                    beam_rank_step<true>(rk[0], rk2[0], ka, kb, k[0], ckey_addr + (b + 8) * 64);
#pragma unroll
                    for (int sl = 1; sl < SL; ++sl) beam_rank_step<false>(rk[sl], rk2[sl], ka, kb, k[sl], 0u);
                } else {
#pragma unroll
                    for (int sl = 0; sl < SL; ++sl) beam_rank_step<false>(rk[sl], rk2[sl], ka, kb, k[sl], 0u);
                    b = nblk;                                                  // done
                }
            }
            if (b < nblk) {
#pragma unroll
                for (int sl = 0; sl < SL; ++sl) beam_rank_step<false>(rk[sl], rk2[sl], kb, ka, k[sl], 0u);
            }
#pragma unroll
            for (int sl = 0; sl < SL; ++sl) atomicAdd(&rank_s[lane + 64 * sl], rk[sl] + rk2[sl]);
        }
        __syncthreads();                                                       // (C)
        int rank = 0;
        if (ci >= 0) { rank = rank_s[slot]; rank_s[slot] = 0; }
        const int4 lv = *reinterpret_cast<const int4 *>(live_s);
        const int nsel = min(beam, lv.x + lv.y + lv.z + lv.w);
        BSTAMP(2)
        // ---- the survivors write next frame's prefixes
        if (key != 0ull && rank < beam) {
            const int R = rank, nl = n_last;
            const float l2 = lpn[nl], lp0n = lpn[0];
            pb[R] = n_pb; tot[R] = n_tot; last[R] = nl; hash[R] = n_hash; phash[R] = n_ph; hx[R] = n_hash * PRIME;
            spb[R] = n_tot + lp0n; spnb[R] = nl > 0 ? n_pnb + l2 : -INFINITY; lpl[R] = nl > 0 ? l2 : 0.f;
            if (bp_in_lds) bpl[t * beam + R] = n_bp; else bpg[(size_t)t * NB + R] = n_bp;
        }
        if (tid >= nsel && tid < beam) {                                       // fewer candidates than the beam holds
            pb[tid] = -INFINITY; tot[tid] = -INFINITY; spb[tid] = -INFINITY; spnb[tid] = -INFINITY; lpl[tid] = 0.f;
            last[tid] = 0; hash[tid] = 0ull; phash[tid] = 0ull; hx[tid] = 0ull;
        }
        BSTAMP(3)
        __syncthreads();                                                       // (A)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- best prefix: walk the back-pointers (one lane), then ends / confidences in parallel over the labels
    int32_t *olab = labels + (size_t)n * max_per_line, *ost = starts + (size_t)n * max_per_line;
    if (bp_in_lds) {
        int32_t *stk = bpl + (size_t)T * beam;                                 // [T][2] (label, frame), last label first
        if (tid == 0) {
            int cnt = 0, e = 0;
            for (int t = len - 1; t >= 0; --t) {
                const int v = bpl[t * beam + e];
                if (v & 0xffff) { stk[2 * cnt] = v & 0xffff; stk[2 * cnt + 1] = t; ++cnt; }
                e = v >> 16;
            }
            s_cnt = cnt;
            counts[n] = cnt;
        }
        __syncthreads();
        const int cnt = s_cnt;
        for (int k2 = tid; k2 < min(cnt, max_per_line); k2 += 256) { olab[k2] = stk[2 * (cnt - 1 - k2)]; ost[k2] = stk[2 * (cnt - 1 - k2) + 1]; }
    } else if (tid == 0) {
        int cnt = 0, e = 0;
        for (int t = len - 1; t >= 0; --t) {
            const int v = bpg[(size_t)t * NB + e];
            if (v & 0xffff) ++cnt;
            e = v >> 16;
        }
        s_cnt = cnt;
        int k2 = cnt;
        e = 0;
        for (int t = len - 1; t >= 0; --t) {
            const int v = bpg[(size_t)t * NB + e];
            if (v & 0xffff) { --k2; if (k2 < max_per_line) { olab[k2] = v & 0xffff; ost[k2] = t; } }
            e = v >> 16;
        }
        counts[n] = cnt;
    }
    __threadfence_block();
    __syncthreads();
    const int cnt = min(s_cnt, max_per_line);
    for (int k2 = tid; k2 < cnt; k2 += 256) {
        const int c = olab[k2], s0 = ost[k2], limit = k2 + 1 < cnt ? ost[k2 + 1] : len;
        int e = s0;
        while (e + 1 < limit && lg[(size_t)(e + 1) * C + c] > lg[(size_t)(e + 1) * C]) ++e;
        float mxp = -INFINITY;
        for (int t = s0; t <= e; ++t) mxp = fmaxf(mxp, lg[(size_t)t * C + c] - *reinterpret_cast<const float *>(rec + (size_t)t * COCR_BEAM_REC + COCR_BEAM_REC_LZ));
        ends[(size_t)n * max_per_line + k2] = e;
        conf[(size_t)n * max_per_line + k2] = expf(mxp);
    }
#ifdef COCR_CHAIN_STAMPS_BUILD
    BSTAMP(4)
    if (dbg && n == 0 && (tid == STAY0 || tid == 0)) for (int k2 = 0; k2 < 5; ++k2) dbg[(tid ? 0 : 5) + k2] = acc_t[k2] + 1;
#endif
}
