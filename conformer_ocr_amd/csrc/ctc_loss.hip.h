// CTC loss and its gradient with respect to the decoder outputs ("probits") -- the loss of the reference's training / validation
// step (model.py:119 `nn.CTCLoss(reduction='sum', zero_infinity=True)`, model.py:136-142 `log_softmax` -> criterion on (T, N, C)).
// Per line n (one workgroup of 4 waves):
//   A  log-softmax of the valid frames (all 4 waves, one frame per wave at a time) -> lp (scratch)
//   B  the forward recursion alpha on wave 0 and the backward recursion beta on wave 1, AT THE SAME TIME (they are independent);
//      the blank-extended label sequence l' (S = 2 L + 1 states) is spread over the lanes, state s on lane s & 63, register s >> 6,
//      so the two neighbours s-1, s-2 (s+1, s+2) of a step arrive by two lane rotations per register and nothing is exchanged through
//      memory inside the T-step dependent chain; the frame's lp values for the next step are requested one step ahead.
//      alpha_t(s) = lse(alpha_{t-1}(s), alpha_{t-1}(s-1), [alpha_{t-1}(s-2) if l'_s != blank and l'_s != l'_{s-2}]) + lp_t(l'_s)
//      nll = -lse(alpha_{len-1}(S-1), alpha_{len-1}(S-2));   beta mirrors it from the last valid frame.
//   C  gradient (all 4 waves, one frame per wave at a time):  d nll / d probits[t, c] = softmax[t, c] - occ[t, c],
//      occ[t, c] = sum_{s: l'_s = c} exp(alpha_t(s) + beta_t(s) - lp_t(c) + nll)   (state posteriors, in [0, 1]).
//      The sum over the states of a class runs in a FIXED order (blank: lane-strided partial sums + the DPP tree; a character: its
//      occurrences chained in label order), so results are reproducible run to run -- no floating-point atomics.
//   Frames t >= len and lines whose nll is infinite (no alignment fits; zero_infinity) get a zero gradient and contribute 0.
// fp32 like torch's kernel for float32 inputs (its CPU result differs from its own float64 result by up to 6e-4 in the gradient at
// T = 300: the log-domain quantities reach 1e3, one ulp there is 1e-4).  Latency-bound by the T-step chain, not by bytes: algorithmic
// bytes = N T C 4 read (+ N T C 4 written for the gradient).
#pragma once
#include "common.hip.h"

#define COCR_CTCL_MAX_LABELS 255        // per line: S = 2 L + 1 <= 512 states = 8 registers of a wave

__device__ __forceinline__ float ctcl_lse3(float a, float b, float c) {
    const float m = fmaxf(a, fmaxf(b, c));
    const float r = m + logf(expf(a - m) + expf(b - m) + expf(c - m));
    return m == -INFINITY ? -INFINITY : r;
}

template <int SJ>
__global__ __launch_bounds__(256) void ctc_loss_kernel(const float *__restrict__ probits, int T, int C, const int32_t *__restrict__ lens,
                                                       const int32_t *__restrict__ label_lens, const int32_t *__restrict__ label_off,
                                                       const int32_t *__restrict__ labels, float *__restrict__ nll_out, float *__restrict__ grad,
                                                       float *__restrict__ lp_all, float *__restrict__ ab_all) {
    constexpr int SP = 64 * SJ;
    extern __shared__ __attribute__((aligned(16))) unsigned char ctcl_smem[];
    int32_t *first = reinterpret_cast<int32_t *>(ctcl_smem);                 // [C] first occurrence of class c in the label sequence, -1 if none
    __shared__ int32_t lab[COCR_CTCL_MAX_LABELS + 1], nxt[COCR_CTCL_MAX_LABELS + 1];
    __shared__ float fin[64 * 8];
    __shared__ float sh_nll;
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int len = min(max(lens[n], 0), T);
    const int L = label_lens[n];
    const int S = 2 * L + 1;
    const float *pr = probits + (size_t)n * T * C;
    float *lp = lp_all + (size_t)n * T * C;
    float *al = ab_all + (size_t)n * 2 * T * SP, *be = al + (size_t)T * SP;
    float *gr = grad ? grad + (size_t)n * T * C : nullptr;

    // ---- labels, occurrence chains
    for (int c = tid; c < C; c += 256) first[c] = -1;
    for (int k = tid; k < L; k += 256) lab[k] = labels[label_off[n] + k];
    __syncthreads();
    for (int k = tid; k < L; k += 256) {
        const int c = lab[k];
        int nx = -1;
        for (int j = k + 1; j < L; ++j) if (lab[j] == c) { nx = j; break; }
        nxt[k] = nx;
        bool is_first = true;
        for (int j = k - 1; j >= 0; --j) if (lab[j] == c) { is_first = false; break; }
        if (is_first) first[c] = k;
    }
    // ---- A: log-softmax of the valid frames
    for (int t = wave; t < len; t += 4) {
        const float *x = pr + (size_t)t * C;
        float mx = -INFINITY;
        for (int c = lane; c < C; c += 64) mx = fmaxf(mx, x[c]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int c = lane; c < C; c += 64) sum += expf(x[c] - mx);
        const float lz = mx + logf(wave_sum(sum));
        for (int c = lane; c < C; c += 64) lp[(size_t)t * C + c] = x[c] - lz;
    }
    __threadfence_block();
    __syncthreads();

    // ---- B: alpha on wave 0, beta on wave 1
    if (wave < 2 && len > 0) {
        const bool fwd = wave == 0;
        int cls[SJ];            // class of state s = lane + 64 j
        bool skip[SJ];          // alpha: s-2 feeds s;  beta: s+2 feeds s
        bool live[SJ];
#pragma unroll
        for (int j = 0; j < SJ; ++j) {
            const int s = lane + 64 * j;
            live[j] = s < S;
            const int k = (s - 1) >> 1;
            cls[j] = (live[j] && (s & 1)) ? lab[k] : 0;
            if (fwd) skip[j] = live[j] && (s & 1) && s >= 3 && lab[k] != lab[k - 1];
            else skip[j] = (s & 1) && s + 2 < S && lab[k] != lab[k + 1];
        }
        float a[SJ], nlp[SJ];
        const int t0 = fwd ? 0 : len - 1, dt = fwd ? 1 : -1;
#pragma unroll
        for (int j = 0; j < SJ; ++j) {
            const int s = lane + 64 * j;
            const float v = live[j] ? lp[(size_t)t0 * C + cls[j]] : -INFINITY;
            const bool init = fwd ? (s < 2) : (s >= S - 2);
            a[j] = (live[j] && init) ? v : -INFINITY;
            (fwd ? al : be)[(size_t)t0 * SP + s] = a[j];
        }
        if (len > 1) {
#pragma unroll
            for (int j = 0; j < SJ; ++j) nlp[j] = live[j] ? lp[(size_t)(t0 + dt) * C + cls[j]] : -INFINITY;
        }
        const int src1 = fwd ? ((lane + 63) & 63) : ((lane + 1) & 63), src2 = fwd ? ((lane + 62) & 63) : ((lane + 2) & 63);
        for (int i = 1; i < len; ++i) {
            const int t = t0 + i * dt;
            float cur[SJ];
#pragma unroll
            for (int j = 0; j < SJ; ++j) cur[j] = nlp[j];
            if (i + 1 < len) {
#pragma unroll
                for (int j = 0; j < SJ; ++j) nlp[j] = live[j] ? lp[(size_t)(t + dt) * C + cls[j]] : -INFINITY;
            }
            float r1[SJ], r2[SJ];
#pragma unroll
            for (int j = 0; j < SJ; ++j) { r1[j] = __shfl(a[j], src1, 64); r2[j] = __shfl(a[j], src2, 64); }
#pragma unroll
            for (int j = 0; j < SJ; ++j) {
                float n1, n2;
                if (fwd) {     // lanes 0 (and 1) take the previous register's lanes 63 (and 62)
                    n1 = lane >= 1 ? r1[j] : (j > 0 ? r1[j > 0 ? j - 1 : 0] : -INFINITY);
                    n2 = lane >= 2 ? r2[j] : (j > 0 ? r2[j > 0 ? j - 1 : 0] : -INFINITY);
                } else {       // lanes 63 (and 62) take the next register's lanes 0 (and 1)
                    n1 = lane < 63 ? r1[j] : (j + 1 < SJ ? r1[j + 1 < SJ ? j + 1 : 0] : -INFINITY);
                    n2 = lane < 62 ? r2[j] : (j + 1 < SJ ? r2[j + 1 < SJ ? j + 1 : 0] : -INFINITY);
                }
                const float v = ctcl_lse3(a[j], n1, skip[j] ? n2 : -INFINITY) + cur[j];
                cur[j] = live[j] ? v : -INFINITY;
            }
#pragma unroll
            for (int j = 0; j < SJ; ++j) {
                a[j] = cur[j];
                (fwd ? al : be)[(size_t)t * SP + lane + 64 * j] = a[j];
            }
        }
        if (fwd) {
#pragma unroll
            for (int j = 0; j < SJ; ++j) fin[lane + 64 * j] = a[j];
        }
    }
    __threadfence_block();
    __syncthreads();
    if (tid == 0) {
        float nll = INFINITY;
        if (len > 0) nll = -ctcl_lse3(fin[S - 1], S > 1 ? fin[S - 2] : -INFINITY, -INFINITY);
        else if (L == 0) nll = 0.f;
        sh_nll = nll;
        nll_out[n] = nll == INFINITY ? 0.f : nll;          // zero_infinity
    }
    __syncthreads();
    if (!gr) return;
    const float nll = sh_nll;
    const int tz = nll == INFINITY ? 0 : len;               // frames from tz on: zero gradient
    // ---- C: gradient
    for (int t = wave; t < tz; t += 4) {
        const float *at = al + (size_t)t * SP, *bt = be + (size_t)t * SP, *lt = lp + (size_t)t * C;
        const float lp0 = lt[0];
        float ob = 0.f;
        for (int k = lane; k <= L; k += 64) ob += expf(at[2 * k] + bt[2 * k] - lp0 + nll);
        ob = wave_sum(ob);
        for (int c = lane; c < C; c += 64) {
            const float l = lt[c];
            float occ = 0.f;
            if (c == 0) occ = ob;
            else for (int k = first[c]; k >= 0; k = nxt[k]) occ += expf(at[2 * k + 1] + bt[2 * k + 1] - l + nll);
            gr[(size_t)t * C + c] = expf(l) - occ;
        }
    }
    for (int i = tz * C + tid; i < T * C; i += 256) gr[i] = 0.f;
}

static inline void launch_ctc_loss(hipStream_t s, int sj, size_t lds, const float *probits, int N, int T, int C, const int32_t *lens, const int32_t *label_lens,
                                   const int32_t *label_off, const int32_t *labels, float *nll, float *grad, float *lp_all, float *ab_all) {
    dim3 grid(N), block(256);
    switch (sj) {
    case 1: hipLaunchKernelGGL((ctc_loss_kernel<1>), grid, block, lds, s, probits, T, C, lens, label_lens, label_off, labels, nll, grad, lp_all, ab_all); break;
    case 2: hipLaunchKernelGGL((ctc_loss_kernel<2>), grid, block, lds, s, probits, T, C, lens, label_lens, label_off, labels, nll, grad, lp_all, ab_all); break;
    case 4: hipLaunchKernelGGL((ctc_loss_kernel<4>), grid, block, lds, s, probits, T, C, lens, label_lens, label_off, labels, nll, grad, lp_all, ab_all); break;
    default: hipLaunchKernelGGL((ctc_loss_kernel<8>), grid, block, lds, s, probits, T, C, lens, label_lens, label_off, labels, nll, grad, lp_all, ab_all); break;
    }
}
