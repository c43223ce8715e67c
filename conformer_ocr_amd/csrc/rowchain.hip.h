// Row-local chains of a conformer block in ONE kernel (bf16; encoder_dim 256 or 512).
//
// Everything in a block except the attention core (needs all frames of a line) maps a row of the (M, D) activation to a row:
// projections, feed-forward modules, residual adds, LayerNorms, and -- with a +-15 frame halo -- the depthwise conv.  A workgroup
// that owns a block of rows therefore runs a whole sequence of them back to back:
//     first chain (block 0):                 frontend output linear (K = F C, + LayerNorm)  ->  FFN 1 (+ residual + LayerNorm)  ->  q/k/v projection
//     chain A (after the attention core):    out-proj + residual + LayerNorm  ->  pointwise conv 1 + GLU
//     chain B (after the GLU):               depthwise conv + BatchNorm + SiLU  ->  pointwise conv 2 + residual + LayerNorm  ->  FFN 2
//                                            (+ block-final LayerNorm + next block's LayerNorm)  ->  next block's FFN 1 (+ LayerNorm)
//                                            ->  its q/k/v projection
// Reference lines: attention.py:70,103 (out_proj), convolution.py:138-143 (pointwise convs, GLU, depthwise conv, BatchNorm, SiLU),
// feed_forward.py:45-52, modules.py:32 (residual), encoder.py:62-99.
//
// Machinery:
//   * 8 waves; the workgroup owns BMC = 16 MT rows (MT = 6: 96 rows, the throughput form; fewer rows = more workgroups for small
//     batches / one batch in flight).  Every product is computed with the WEIGHTS on the MFMA row side: accumulator register q of
//     lane (r16 = lane & 15, g = lane >> 4) of tile (i, j) is out[row 16 i + r16][column 32 wave + 16 j + 4 g + q] of a 256-column
//     step, i.e. a lane holds 4 consecutive output columns of one activation row.
//   * weights stream L2 -> REGISTERS from a fragment-major copy (pack_frag_kernel: [n/32][k/32][2][64 lanes][8]): the 16 B-fragments
//     of one step (32 output columns x 256 k) are one contiguous 16 KiB run per wave, requested one step ahead through a 16-fragment
//     register ring (the compiler's vmcnt bookkeeping orders it).
//   * activations are MFMA operands read from swizzled LDS images ([k/64 panels][rows][128 B]): the D-wide operand rows and, for the
//     FFN, two 256-wide hidden-chunk images (software pipeline: bias + SiLU of chunk c folded into the k-steps of the second
//     product of chunk c-1).
//   * THE RESIDUAL STREAM LIVES IN THE ACCUMULATORS: x (fp32) is loaded once per launch in the accumulator layout and is the C operand
//     of every stage's product (x + bias as the initial accumulator; the half-step factor of the FFN is folded into the packed copy of
//     its second matrix), so the residual add costs nothing and x never round-trips through memory inside a launch.
//   * LayerNorm straight from those registers: per-lane partial sums (8 or 16 values per row) -> two lane-swap butterflies
//     (v_permlane32_swap / v_permlane16_swap) -> per-wave partials in LDS -> one barrier -> every lane sums the 8 waves' partials of
//     its rows; the normalised values go straight into the operand image as bf16.  (The first version staged the fp32 tile in LDS,
//     re-read x from global in a row-per-16-lanes layout and wrote x back after every stage: 3 round trips of the stream per launch.)
#pragma once
#include <type_traits>

#include "gemm.hip.h"
#include "rowchain_args.hip.h"

#ifndef COCR_STAMP_WG
#define COCR_STAMP_WG 7        // dev (stamps build): the workgroup whose phase boundaries are stamped
#endif
#ifndef COCR_RC_EXP
#define COCR_RC_EXP 0          // dev: timing experiments (wrong results): 1 no stream load, 2 no stream store, 4 no partial read-back, 8 no depthwise FMAs, 16 no SiLU transcendentals,
                               // 32 no LayerNorm statistics, 64 no matrix instructions, 128 no weight stream (the ring is loaded once), 256 one operand fragment set per step,
                               // 512 no barriers, 1024 no q/k/v staging and copy-out, 2048 no depthwise prologue arithmetic at all, 4096 no SiLU tiles, 8192 no normalise, 16384 no SiLU in the depthwise prologue
#endif

// Lane-swap butterflies (gfx950).  v_permlane32_swap a, b: a <- [a.lo32, b.lo32], b <- [a.hi32, b.hi32]; v_permlane16_swap a, b (rows of 16
// lanes r0..r3): a <- [a.r0, b.r0, a.r2, b.r2], b <- [a.r1, b.r1, a.r3, b.r3].  Inline asm: the builtin's two results come back as ONE register when
// both are consumed by a floating-point add (hipcc 7.2: `v_add_f32 v1, v1, v1`); the s_nop covers the VALU-write -> permlane-read hazard.
__device__ __forceinline__ void swap32(float &a, float &b) { asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap16(float &a, float &b) { asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
// four per-lane values a, b, c, d, each to be summed over the lanes {l, l^16, l^32, l^48}: returns ONE register whose 16-lane row rho holds
// the total of {a, c, b, d}[rho]
__device__ __forceinline__ float butterfly4(float a, float b, float c, float d) {
    swap32(a, b);
    swap32(c, d);
    float ab = a + b, cd = c + d;          // ab: [sum over halves of a | of b]
    swap16(ab, cd);
    return ab + cd;                        // rows: a, c, b, d
}

// hardware workgroup id -> row block such that XCD (id % 8) owns a contiguous range of row blocks (as attention.hip.h: xcd_remap_i)
__device__ __forceinline__ int xcd_remap_rows(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// Dev build -DCOCR_RC_PAIR=1 (round 4; measured, NOT the default -- DESIGN section 4b): the 48-row form of the 256-wide
// engine built to be CO-RESIDENT, two workgroups per CU (4 waves per SIMD: <= 128 registers, 80 512 B of LDS each), so that one workgroup's
// prologue / epilogues / barriers run beside the other's k-steps.  Its weight ring holds 8 fragments (half a step ahead) instead of 16 and
// its depthwise prologue takes 128 channels per pass.  Bit-identical to the other forms.  At 2 x 250 workgroups on 250 CUs a launch takes
// 98 us against 77 us for 250 96-row workgroups: two 48-row workgroups stream every weight byte twice through the CU's L1 path, and that
// costs more than the overlap returns.
#ifndef COCR_RC_PAIR
#define COCR_RC_PAIR 0
#endif
template <int D, int MT> constexpr bool rowchain_pair_form() { return COCR_RC_PAIR && D == 256 && MT == 3; }

// slack behind the two hidden-chunk images (the q / k / v output tiles' padded rows reach into it)
template <int D, int MT> constexpr size_t rowchain_slack_bytes() { return rowchain_pair_form<D, MT>() ? 2048 : 4096; }
template <int D, int MT> constexpr size_t rowchain_lds_bytes() {
    return (size_t)(D / 64) * 16 * MT * 128 + 2 * 4 * (size_t)(16 * MT) * 128 + rowchain_slack_bytes<D, MT>() + 16 * (size_t)D + 12 * (size_t)(16 * MT) + 64;
}
// channels per pass of the depthwise prologue (window + taps of that many channels in LDS at a time): the co-resident form takes 128
template <int D, int MT> constexpr int rowchain_dw_pass() { return rowchain_pair_form<D, MT>() ? 128 : 256; }
// Where the depthwise taps of a channel pass ((DWK + 1) rows of floats) are staged in LDS: 1 = behind the window inside the hidden-image
// area, 2 = in an area of their own behind everything else, 0 = nowhere (no room: every thread loads its own taps from global memory)
template <int D, int MT, int DWK> constexpr int rowchain_taps_place() {
    if (DWK == 0 || (DWK + 1) % 8 != 0) return 0;
    constexpr size_t cw = rowchain_dw_pass<D, MT>(), rpi = 64 / (cw / 8);
    constexpr size_t hsb = 2 * 4 * (size_t)(16 * MT) * 128 + rowchain_slack_bytes<D, MT>(), win = ((16 * MT + DWK - 1 + rpi - 1) / rpi) * 1024, taps = (size_t)(DWK + 1) * cw * 4;
    if (win + taps <= hsb) return 1;
    if (rowchain_pair_form<D, MT>()) return 0;
    if (rowchain_lds_bytes<D, MT>() + taps <= 160 * 1024) return 2;
    return 0;
}

// KD / KL: zero-padded narrow models (cocr_api.hip: set_engine_dims) -- the k-steps (32 k each) of a K = D product and of the FFN's LAST
// hidden chunk that hold real columns; the k-steps beyond them multiply zeros and are skipped (their weight fragments are still streamed:
// the ring's bookkeeping stays one shape).  8 / 8 = nothing skipped.  The reference's default model (encoder_dim 144, feed-forward 576 in
// a 256 / 768-wide engine): 5 / 2.
template <int D, int MT, int DWK, int K0, int K1, int K2, int K3, bool TAPS, int KD = 8, int KL = 8>
__global__ __launch_bounds__(512, (rowchain_pair_form<D, MT>() ? 4 : 2)) void rowchain_kernel(ChainArgs p) {
    typedef bf16_t T;
    static_assert(D == 256 || D == 512, "encoder_dim of the row-chain kernels");
    static_assert(MT >= 2 && MT <= 6, "16-row tiles per workgroup");
    static_assert(KD >= 1 && KD <= 8 && KL >= 1 && KL <= 8 && (D == 256 || (KD == 8 && KL == 8)), "skipped k-steps: the 256-wide engine only");
    typedef std::integral_constant<int, 8> KK8;
    typedef std::integral_constant<int, KD> KKD;           // K = D products
    typedef std::integral_constant<int, KL> KKL;           // the second product of the FFN's last hidden chunk
    typedef KKD KKQ;
    typedef KKD KKG;
    typedef KKD KKP1;
    // the ROWLN stage of the out-proj -> GLU chain multiplies the attention context, whose real columns sit in head slots all over the
    // padded width ([64 h, 64 h + d_head)): no k-step of it is all zeros
    typedef std::integral_constant<int, (K1 == ST_GLU) ? 8 : KD> KKR;
    constexpr int BMC = 16 * MT;
    constexpr int KS = D / 256;                // 256-deep k slices of a K = D product
    constexpr int NS = D / 256;                // 256-column steps of an N = D product
    constexpr int NJ = 2 * NS;                 // 16-column tiles of this wave per row tile of an N = D product
    constexpr int PANEL = BMC * 128;           // one [rows][128 B] panel of an operand image (64 bf16 of k per row)
    constexpr int IMGX = (D / 64) * PANEL;     // BMC x D bf16
    constexpr int IMGH = 4 * PANEL;            // BMC x 256 bf16 (hidden chunk)
    constexpr int OS = 256 * 2 + 16;           // bf16 staged row of a 256-column output tile (q / k / v)
    constexpr int OSD = D * 2 + 16;            // bf16 staged row of a D-column output tile (GLU)
    constexpr int SLICE = 16 * 512;            // elements in one step's weight run (16 fragments)
    constexpr int PROW = 80;                   // LayerNorm partials: 8 waves x (sum, sum of squares) per row, padded 64 -> 80 bytes (the 16 rows a
                                               // ds_read_b128 lane group touches then fall on 16 different bank quads)
    constexpr int HSB = 2 * IMGH + (int)rowchain_slack_bytes<D, MT>();       // hidden images + slack (output tiles, depthwise window, LayerNorm partials alias them)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *xa = smem;                  // operand image
    unsigned char *hs = smem + IMGX;           // 2 hidden-chunk images
    float *lnp = reinterpret_cast<float *>(smem + IMGX + HSB);                       // LayerNorm parameters of the running stage: g1, b1, g2, b2 (4 x D floats)
    long long *rowoff = reinterpret_cast<long long *>(smem + IMGX + HSB + 16 * D);   // row -> offset of its (line, frame) in the q / k / v layouts
    int *tpos = reinterpret_cast<int *>(smem + IMGX + HSB + 16 * D + 8 * BMC);      // frame index of each row inside its line

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4, swz = r16 & 7;
    // XCD-aware order (p.xcd_order): hardware workgroup ids go round the 8 XCDs; the row block a workgroup owns is chosen so that each XCD
    // holds a contiguous range of row blocks, i.e. of lines -- the same lines whose (line, head) workgroups the attention kernels put on that
    // XCD: what one launch writes (context rows, GLU output, stream, q / k / v) the next one reads from the same L2
    const int rblk = p.xcd_order ? xcd_remap_rows(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int M = p.M, m0 = rblk * BMC, mend = min(M, m0 + BMC);
    const int lrow = lane >> 3, cpos = lane & 7;
#ifdef COCR_CHAIN_STAMPS_BUILD
    int sk = 0;
#define RSTAMP() { if (p.stamps && rblk == COCR_STAMP_WG && lane == 0) p.stamps[wave * 64 + sk] = __builtin_readcyclecounter(); ++sk; }
#else
#define RSTAMP()
#endif
    RSTAMP()                                       // 0: start
    auto lds_fence_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (!(COCR_RC_EXP & 512)) __builtin_amdgcn_s_barrier();
    };

    constexpr int DWPAD = DWK ? (DWK - 1) / 2 : 0, DWROWS = BMC + 2 * DWPAD;
    constexpr int CW = rowchain_dw_pass<D, MT>();          // channels per pass of the depthwise prologue
    constexpr int WS = CW * 2;                 // bytes per window row
    constexpr int LPR = CW / 8, RPI = 64 / LPR;            // lanes per window row (16 B each), window rows per wave-instruction
    constexpr int WINB = ((DWROWS + RPI - 1) / RPI) * 1024;
    static_assert(DWK == 0 || (size_t)WINB <= (size_t)HSB, "depthwise window (CW channels at a time) must fit the hidden-image area");
    static_assert((size_t)BMC * OSD <= (size_t)HSB && 2 * (size_t)BMC * OS <= (size_t)HSB, "output tiles alias the hidden-image area");
    // depthwise window of channel pass `h`: rows m0 - PAD .. m0 + BMC + PAD - 1 of the GLU output (addresses clamped; frames outside the row's
    // own line are excluded by the tap range below), [row][WS bytes] at hs: one wave-instruction = RPI rows
    auto dw_window = [&](int h) {
        for (int q2 = wave; q2 < WINB / 1024; q2 += 8) {
            const int j = RPI * q2 + lane / LPR, mrow = min(max(m0 - DWPAD + j, 0), M - 1);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(p.dw_in + (size_t)mrow * D + h * CW + (lane % LPR) * 8), (lds_ptr_t)(hs + q2 * 1024), 16, 0, 0);
        }
    };
    // FRONT (first stage = the frontend's output linear, K = F C): its (BMC x 256)-deep operand slices go through the two hidden-image areas,
    // double buffered, MT wave-instructions per wave and slice (4 panels x BMC / 8 row groups of 8 rows x 128 B)
    constexpr bool FRONT0 = K0 == ST_FRONT;
    static_assert(!FRONT0 || DWK == 0, "the output linear starts a launch");
    static_assert(K1 != ST_FRONT && K2 != ST_FRONT && K3 != ST_FRONT, "FRONT is a first stage");
    auto front_dma = [&](int ks) {
        unsigned char *buf = hs + (ks & 1) * IMGH;
        const int Kf = p.st[0].K;
#pragma unroll
        for (int u = 0; u < MT; ++u) {
            const int id = wave + 8 * u, pnl = id / (BMC / 8), rg = id - pnl * (BMC / 8), row = rg * 8 + lrow;
            const T *src = p.A0 + (size_t)min(m0 + row, M - 1) * Kf + ks * 256 + pnl * 64 + ((cpos ^ (row & 7)) * 8);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(buf + pnl * PANEL + rg * 1024), 16, 0, 0);
        }
    };
    static_assert(4 * (BMC / 8) == 8 * MT, "operand slice: MT wave-instructions per wave");
    if constexpr (FRONT0) {
        front_dma(0);
    } else if constexpr (DWK == 0) {
        // ---- first operand tile -> LDS image: (D / 64) panels x BMC / 8 row groups of 8 rows, one wave-instruction each
        for (int id = wave; id < (D / 64) * (BMC / 8); id += 8) {
            const int pnl = id / (BMC / 8), rg = id - pnl * (BMC / 8), row = rg * 8 + lrow;
            const T *src = p.A0 + (size_t)min(m0 + row, M - 1) * D + pnl * 64 + ((cpos ^ (row & 7)) * 8);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(xa + pnl * PANEL + rg * 1024), 16, 0, 0);
        }
    } else {
        dw_window(0);
    }
    // depthwise taps and bias of a channel half: DWK + 1 rows of 256 floats behind the window in LDS, one DMA each, requested with the window
    // and ahead of the weight ring (loads return in order: behind the ring they would wait for all of it).  Every thread then reads its
    // channel pair's taps from there.  (Each thread loading its own taps from global memory -- 32 eight-byte loads, the same 32 KB four times
    // per workgroup -- made 68 loads per wave at the kernel's start, more than the 63 a wave can have outstanding: the prologue started
    // 16 k cycles into the launch.)
    typedef float dw_f32x2 __attribute__((ext_vector_type(2)));
    dw_f32x2 dw_wt[DWK ? DWK : 1], dw_bias = {0.f, 0.f};
    constexpr int TAPLACE = rowchain_taps_place<D, MT, DWK>();
    unsigned char *tapa = TAPLACE == 1 ? hs + WINB : smem + rowchain_lds_bytes<D, MT>();
    constexpr int TPI = 256 / CW;                  // tap rows (CW floats) per wave-instruction
    auto dw_taps = [&](int h) {
        if constexpr (DWK != 0 && TAPLACE != 0) {
#pragma unroll
            for (int r2 = wave; r2 < (DWK + 1) / TPI; r2 += 8) {           // (DWK + 1) % 8 == 0: the same number of requests in every wave
                const int r = r2 * TPI + lane / (64 / TPI);
                const float *src = (r < DWK ? p.dw_w + (size_t)r * D : p.dw_b) + h * CW + (lane % (64 / TPI)) * 4;
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(tapa + r2 * 1024), 16, 0, 0);
            }
        } else if constexpr (DWK != 0) {
            const int c = h * CW + 2 * (tid % (CW / 2));
#pragma unroll
            for (int tau = 0; tau < DWK; ++tau) dw_wt[tau] = *reinterpret_cast<const dw_f32x2 *>(p.dw_w + (size_t)tau * D + c);
            dw_bias = *reinterpret_cast<const dw_f32x2 *>(p.dw_b + c);
        }
    };
    auto dw_taps_read = [&]() {                    // after the barrier that publishes the window and the taps
        if constexpr (DWK != 0 && TAPLACE != 0) {
            const unsigned char *tb = tapa + (tid % (CW / 2)) * 8;
#pragma unroll
            for (int tau = 0; tau < DWK; ++tau) dw_wt[tau] = *reinterpret_cast<const dw_f32x2 *>(tb + tau * (CW * 4));
            dw_bias = *reinterpret_cast<const dw_f32x2 *>(tb + DWK * (CW * 4));
        }
    };
    RSTAMP()                                       // window / operand tile requested
    dw_taps(0);
    RSTAMP()                                       // taps requested
    // ---- weight ring: the first step's 16 fragments
    // RING = 16: the ring holds the current step's 16 fragments, slot f refilled with fragment f of the next step's run as soon as its
    // k-step is done.  RING = 8 (the co-resident form): slot f % 8 holds fragment f; after k-step kk < 4 its two slots take fragments
    // 2 kk + 8 (+ 1) of the CURRENT run, after k-step kk >= 4 fragments 2 (kk - 4) (+ 1) of the next one -- half a step ahead.
    constexpr int RING = rowchain_pair_form<D, MT>() ? 8 : 16;
    bf16x8 ring[RING];
    auto fill = [&](const T *slice, int f) { ring[f % RING] = *reinterpret_cast<const bf16x8 *>(slice + f * 512 + lane * 8); };
    const T *cur_run = p.st[0].W + (size_t)wave * (FRONT0 ? p.st[0].K / 256 : KS) * SLICE;      // the run whose fragments the ring holds (RING = 8)
    {
#pragma unroll
        for (int f = 0; f < RING; ++f) fill(cur_run, f);
    }
    __builtin_amdgcn_sched_barrier(0);
    RSTAMP()                                       // ring requested
    // ---- the residual stream of this workgroup's rows, in the accumulator layout: xs[i][2 ns + j][q] = x[m0 + 16 i + r16][256 ns + 32 wave + 16 j + 4 g + q].
    // Requested AFTER the operand tile / depthwise window and the ring (loads return in order): the prologue and the first product do not
    // wait for these 16 MT D bytes per wave (a CU takes in ~10 B per cycle: 96 KiB = 4.6 us), only the first epilogue does.
    // With the depthwise prologue they are requested after it: their 8 MT NJ registers are free for the prologue (with them the ring did
    // not fit: fragments were parked in scratch memory at the kernel's start, i.e. waited for), and the first product hides them.
    f32x4 xs[MT][NJ];
    // (blocked form: see ChainArgs::x_in_blocked)
    const size_t xblk = ((size_t)rblk * 8 + wave) * (MT * NJ) * 256 + lane * 4;      // floats
    auto load_stream = [&]() {
        if (p.x_in_blocked) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int c = 0; c < NJ; ++c) {
                    if constexpr (COCR_RC_EXP & 1) xs[i][c] = (f32x4){(float)r16, 1.f, (float)g, 0.5f};
                    else xs[i][c] = *reinterpret_cast<const f32x4 *>(p.x + xblk + (i * NJ + c) * 256);
                }
            return;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const float *xrow = p.x + (size_t)min(m0 + 16 * i + r16, mend - 1) * D + 32 * wave + 4 * g;
#pragma unroll
            for (int c = 0; c < NJ; ++c) {
                if constexpr (COCR_RC_EXP & 1) xs[i][c] = (f32x4){(float)r16, 1.f, (float)g, 0.5f};
                else xs[i][c] = *reinterpret_cast<const f32x4 *>(xrow + 256 * (c >> 1) + 16 * (c & 1));
            }
        }
    };
    constexpr int EARLY_XS = (DWK == 0 && !FRONT0) ? MT * NJ : 0;     // stream loads in flight at the first wait
    if constexpr (DWK == 0 && !FRONT0) load_stream();                 // (FRONT: the stream is born in this launch)
    __builtin_amdgcn_sched_barrier(0);
    if (tid < BMC) {
        const int m = min(m0 + tid, M - 1), b = m / p.T_, t = m - b * p.T_;
        rowoff[tid] = ((long long)b * p.heads * p.Tp + t) * p.dhp;
        tpos[tid] = t;
    }
    RSTAMP()                                       // 1: loads requested
    // the operand DMAs (older than the ring and stream loads) have landed.  The count is the FEWEST loads that can be in flight behind them:
    // a first step that skips k-steps never reads its last ring slots and the compiler drops those loads (with 16 here and 10 loads issued the
    // wait let the depthwise window arrive late: wrong rows in the first launch on fresh LDS, right ones by luck afterwards)
    constexpr int RING0 = RING == 16 ? 2 * (K0 == ST_ROWLN ? KKR::value : K0 == ST_FFN ? KKP1::value : 8)
                                     : (2 * (K0 == ST_ROWLN ? KKR::value : K0 == ST_FFN ? KKP1::value : 8) < 8 ? 2 * (K0 == ST_ROWLN ? KKR::value : K0 == ST_FFN ? KKP1::value : 8) : 8);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(RING0 + EARLY_XS) : "memory");
    lds_fence_barrier();
    RSTAMP()                                       // 2: operand tile / window landed

    if constexpr (DWK != 0) {
        // Depthwise conv (kernel DWK, zero padding at the line ends, BatchNorm folded) + SiLU on the GLU output (convolution.py:140-142),
        // for the workgroup's rows from the (BMC + DWK - 1)-row window in LDS, written straight into the operand image; 256 channels at a
        // time.  thread = one channel pair x BMC / 4 rows (groups of 8); a wave's lanes share their rows, so the boundary test is uniform.
        // Accumulation order per output: bias, then taps ascending (as the stand-alone kernel: bit-identical).
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        constexpr int CPAIRS = CW / 2, NRG = 512 / CPAIRS;    // channel pairs of a pass; row groups the threads split the block's rows into
        constexpr int RQ = BMC / NRG;                         // rows of one thread
        constexpr int G = RQ % 8 == 0 ? 8 : (RQ % 6 == 0 ? 6 : 8);      // rows walked together (a window row read serves G outputs)
        static_assert(BMC % NRG == 0 && RQ >= G, "rows per thread of the depthwise prologue");
        const int cp = tid % CPAIRS, rq = __builtin_amdgcn_readfirstlane(tid / CPAIRS);
        const int T_ = p.T_;
#pragma unroll 1
        for (int h = 0; h < D / CW; ++h) {
            if (h > 0) {                                     // next channel pass: the window area is free after the barrier that ended the previous one
                dw_window(h);
                dw_taps(h);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                lds_fence_barrier();
            }
            const int c = 2 * cp;                            // channel inside the pass
            dw_taps_read();
            const f32x2_t bias = dw_bias;
            // rows of this thread: group rq of the block, G at a time (a ragged last group re-does rows of the previous one)
#pragma unroll 1
            for (int r0 = RQ * rq; r0 < ((COCR_RC_EXP & 2048) ? RQ * rq : RQ * (rq + 1)); r0 += G) {
                const int rb = min(r0, RQ * (rq + 1) - G);
                const int t0 = __builtin_amdgcn_readfirstlane(tpos[rb]);
                f32x2_t acc[G];
#pragma unroll
                for (int i = 0; i < G; ++i) acc[i] = bias;
                const unsigned char *wbase = hs + (size_t)rb * WS + c * 2;
                if (t0 >= DWPAD && t0 + (G - 1) + DWPAD < T_) {        // all G rows inside one line, full tap range
#pragma unroll
                    for (int rin = 0; rin < G + DWK - 1; ++rin) {
                        const bf16x2 xv = *reinterpret_cast<const bf16x2 *>(wbase + rin * WS);
                        const f32x2_t xf = {(float)xv[0], (float)xv[1]};
#pragma unroll
                        for (int i = 0; i < G; ++i) {
                            const int tau = rin - i;
                            if (tau >= 0 && tau < DWK) { if constexpr (COCR_RC_EXP & 8) { if (tau == 0) acc[i] += xf; } else acc[i] = __builtin_elementwise_fma(dw_wt[tau], xf, acc[i]); }
                        }
                    }
                } else if (__builtin_amdgcn_readfirstlane(tpos[rb + G - 1]) == t0 + G - 1) {
                    // near a line end, the G rows inside one line: the same walk over the window with the rows outside the line read as zero
                    // (a zero tap product leaves the sum as it is: the same values as skipping the tap).  The per-tap form below costs 3 x
                    // the instructions; with one line end per three 96-row blocks it set the kernel's duration: the launch is as long as its
                    // slowest workgroup (timing experiment without the prologue's arithmetic: - 11 us of 63).
#pragma unroll
                    for (int rin = 0; rin < G + DWK - 1; ++rin) {
                        const bool ok = (unsigned)(t0 + rin - DWPAD) < (unsigned)T_;     // (uniform)
                        const bf16x2 xv = *reinterpret_cast<const bf16x2 *>(wbase + rin * WS);
                        const f32x2_t xf = {ok ? (float)xv[0] : 0.f, ok ? (float)xv[1] : 0.f};
#pragma unroll
                        for (int i = 0; i < G; ++i) {
                            const int tau = rin - i;
                            if (tau >= 0 && tau < DWK) acc[i] = __builtin_elementwise_fma(dw_wt[tau], xf, acc[i]);
                        }
                    }
                } else {                                               // rows of two lines: tap tau of row i is in range iff 0 <= t_i + tau - PAD < T
                    int ti[G];
#pragma unroll
                    for (int i = 0; i < G; ++i) ti[i] = __builtin_amdgcn_readfirstlane(tpos[rb + i]);
#pragma unroll
                    for (int rin = 0; rin < G + DWK - 1; ++rin) {      // the same walk, the window row masked per output row
                        const bf16x2 xv = *reinterpret_cast<const bf16x2 *>(wbase + rin * WS);
                        const f32x2_t xr = {(float)xv[0], (float)xv[1]};
#pragma unroll
                        for (int i = 0; i < G; ++i) {
                            const int tau = rin - i;
                            if (tau >= 0 && tau < DWK) {
                                const bool ok = (unsigned)(ti[i] + tau - DWPAD) < (unsigned)T_;
                                const f32x2_t xf = {ok ? xr[0] : 0.f, ok ? xr[1] : 0.f};
                                acc[i] = __builtin_elementwise_fma(dw_wt[tau], xf, acc[i]);
                            }
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    const int row = rb + i, cc = h * CW + c;
                    const bf16x2 o = (COCR_RC_EXP & 16384) ? (bf16x2){(T)acc[i][0], (T)acc[i][1]} : (bf16x2){(T)silu_f(acc[i][0]), (T)silu_f(acc[i][1])};
                    *reinterpret_cast<bf16x2 *>(xa + (cc >> 6) * PANEL + row * 128 + ((((cc & 63) >> 3) ^ (row & 7)) << 4) + (cc & 7) * 2) = o;
                    if constexpr (TAPS) {
                        if (p.tap_dw && m0 + row < mend) *reinterpret_cast<bf16x2 *>(p.tap_dw + (size_t)(m0 + row) * D + cc) = o;
                    }
                }
            }
            lds_fence_barrier();                             // this pass of the operand image complete; the window is free
        }
        load_stream();
        __builtin_amdgcn_sched_barrier(0);
        RSTAMP()                                   // 3: depthwise prologue done
    }

    // One step: acc[rows][32 columns of this wave] += image[256 k] . ring ; ring <- the 16 fragments at `nxt`.  `side(kk)` is independent VALU
    // work folded into the k-step.  `fresh`: the accumulators start from zero -- the first k-step's MFMAs take the constant 0 as their C
    // operand instead of registers zeroed by v_mov (the VALU is the scarce unit of this kernel).
    auto step = [&](const unsigned char *img, auto &acc, int c0, const T *nxt, auto &&side, auto FRESH, auto KKC) {      // acc[MT][..]: tiles c0, c0 + 1
        constexpr bool fresh = decltype(FRESH)::value;
        constexpr int KK = decltype(KKC)::value;           // k-steps with real columns (the others multiply zeros: skipped)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            // row tiles in two halves: half the operand registers live
            constexpr int HT = MT > 3 ? MT / 2 : MT;
#pragma unroll
            for (int h0 = 0; h0 < MT; h0 += HT) {
                bf16x8 a[HT];
#pragma unroll
                for (int i = 0; i < HT; ++i) {
                    if constexpr (COCR_RC_EXP & 256) { if (kk == 0) a[i] = lds_frag_swz(img + (16 * (h0 + i) + r16) * 128, 0, g, swz, T()); else asm volatile("" : "+v"(a[i])); }
                    else a[i] = lds_frag_swz(img + (kk >> 1) * PANEL + (16 * (h0 + i) + r16) * 128, kk & 1, g, swz, T());
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < HT; ++i) {
                        if constexpr (COCR_RC_EXP & 64) {
                            if (fresh && kk == 0) acc[h0 + i][c0 + j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                            asm volatile("" : "+v"(acc[h0 + i][c0 + j]) : "v"(ring[2 * kk + j]), "v"(a[i]));
                        } else acc[h0 + i][c0 + j] = mma16(ring[(2 * kk + j) % RING], a[i], (fresh && kk == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[h0 + i][c0 + j]);
                    }
            }
            if constexpr (!(COCR_RC_EXP & 128)) {
                if constexpr (RING == 16) {
                    fill(nxt, 2 * kk);
                    fill(nxt, 2 * kk + 1);
                } else if (kk < 4 && kk + 4 < KK) {          // (compile-time: kk is an unrolled constant) this slot is read again at k-step kk + 4
                    fill(cur_run, 2 * kk + 8);
                    fill(cur_run, 2 * kk + 9);
                } else {
                    fill(nxt, 2 * (kk & 3));
                    fill(nxt, 2 * (kk & 3) + 1);
                }
            }
            side(kk);
            __builtin_amdgcn_sched_barrier(0);               // keep the refill (and the side work) here: the scheduler otherwise sinks all of it to the step's end
        }
        if constexpr (KK < (RING == 16 ? 8 : 4) && !(COCR_RC_EXP & 128)) {
#pragma unroll
            for (int f = 2 * KK; f < RING; ++f) fill(nxt, f);  // the skipped k-steps' ring slots: the next step's fragments all the same
            __builtin_amdgcn_sched_barrier(0);
        }
        cur_run = nxt;
    };
    auto no_side = [](int) {};
    // LayerNorm parameters of a stage -> LDS, requested at the stage's start by waves 0..3 (one array each); the epilogue's vmcnt(16) + barrier
    // publishes them (at least one step = 16 younger ring loads lies between)
    auto request_ln_params = [&](const ChainStage &st) {
        if (wave < 4) {
            const float *src = wave == 0 ? st.g1 : wave == 1 ? st.b1 : wave == 2 ? (st.g2 ? st.g2 : st.g1) : (st.b2 ? st.b2 : st.b1);
#pragma unroll
            for (int c = 0; c < D / 256; ++c)
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + c * 256 + lane * 4), (lds_ptr_t)(reinterpret_cast<unsigned char *>(lnp) + wave * D * 4 + c * 1024), 16, 0, 0);
        }
    };
    // this wave's 2 x 4 bias values of a 256-column step (requested BEFORE the step: younger loads than the ring's would make their
    // consumer wait for the whole ring)
    struct Bias2 { f32x4 v[2]; };
    auto load_bias = [&](const float *bias) {
        Bias2 b;
        b.v[0] = *reinterpret_cast<const f32x4 *>(bias + 32 * wave + 4 * g);
        b.v[1] = *reinterpret_cast<const f32x4 *>(bias + 32 * wave + 16 + 4 * g);
        return b;
    };
    auto add_bias_to_stream = [&](const float *bias, float alpha) {       // xs += alpha * bias: the stream becomes the product's initial accumulator
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) {
            const Bias2 b = load_bias(bias + 256 * ns);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < MT; ++i) xs[i][2 * ns + j] += b.v[j] * alpha;
        }
    };
    // fp32 stream -> global in the accumulator layout (16 bytes per lane, 64-byte row segments; once per launch and consumer)
    auto store_stream = [&](float *dst) {
        if constexpr (COCR_RC_EXP & 2) { asm volatile("" :: "v"(xs[0][0][0])); return; }
        if (dst == p.x && p.x_out_blocked) {                  // (rows beyond M inside the last block carry finite values nobody uses)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int c = 0; c < NJ; ++c) *reinterpret_cast<f32x4 *>(dst + xblk + (i * NJ + c) * 256) = xs[i][c];
            return;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if (m0 + 16 * i + r16 < mend) {
                float *xrow = dst + (size_t)(m0 + 16 * i + r16) * D + 32 * wave + 4 * g;
#pragma unroll
                for (int c = 0; c < NJ; ++c) *reinterpret_cast<f32x4 *>(xrow + 256 * (c >> 1) + 16 * (c & 1)) = xs[i][c];
            }
        }
    };
    // Row statistics of the stream registers: per-lane sums of its NJ x 4 values per row tile, butterflies over the 4 lanes of a row, per-wave
    // partials (sum, sum of squares) in LDS at `pbuf` [row][wave], one barrier, then every lane adds the 8 waves' partials of its MT rows.
    // One pass (E[x^2] - mean^2 in fp32 over D <= 512 values of magnitude ~1: the bf16 operand this feeds has 8 significant bits).
    auto row_stats = [&](unsigned char *pbuf, bool wait_params, float (&mean)[MT], float (&rstd)[MT]) {
        if constexpr (COCR_RC_EXP & 32) { for (int i = 0; i < MT; ++i) { mean[i] = 0.f; rstd[i] = 1.f; } lds_fence_barrier(); return; }
        float s[MT], ss[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            // (explicit fused multiply-adds here and in `normalise`: every instantiation -- rows per workgroup, debug taps -- must round a row's
            // statistics the same way, whatever the compiler's contraction choices are in that instantiation)
            f32x4 a = xs[i][0], b = xs[i][0] * xs[i][0];
#pragma unroll
            for (int c = 1; c < NJ; ++c) { a += xs[i][c]; b = __builtin_elementwise_fma(xs[i][c], xs[i][c], b); }
            s[i] = (a[0] + a[1]) + (a[2] + a[3]);
            ss[i] = (b[0] + b[1]) + (b[2] + b[3]);
        }
#pragma unroll
        for (int i0 = 0; i0 < MT; i0 += 4) {
            const int i1 = min(i0 + 1, MT - 1), i2 = min(i0 + 2, MT - 1), i3 = min(i0 + 3, MT - 1);
            const float ts = butterfly4(s[i0], s[i1], s[i2], s[i3]), tss = butterfly4(ss[i0], ss[i1], ss[i2], ss[i3]);
            const int tile = g == 0 ? i0 : g == 1 ? i2 : g == 2 ? i1 : i3;              // which tile's totals this lane's row holds
            *reinterpret_cast<f32x2 *>(pbuf + (16 * tile + r16) * PROW + wave * 8) = (f32x2){ts, tss};
        }
        if (wait_params && wave < 4) {      // this stage's LayerNorm parameters have landed (requested before the stage's first step)
            if constexpr (KD == 8 && KL == 8) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(RING) : "memory");      // ... at least one step = 16 younger ring loads lies between (8 of them in flight at most with the short ring)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                    // (steps that skip k-steps issue fewer: the compiler drops the dead ones)
        }
        lds_fence_barrier();
        const float inv_d = p.inv_d;                       // (raw moments: zero-padded columns add nothing, the divisor is the real width)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const unsigned char *pr = pbuf + (16 * i + r16) * PROW;
            f32x4 v = *reinterpret_cast<const f32x4 *>(pr);
            if constexpr (!(COCR_RC_EXP & 4)) {
#pragma unroll
                for (int w2 = 1; w2 < 4; ++w2) v += *reinterpret_cast<const f32x4 *>(pr + 16 * w2);
            }
            const float S = v[0] + v[2], SS = v[1] + v[3];
            mean[i] = S * inv_d;
            const float ex2 = SS * inv_d;
            rstd[i] = __builtin_amdgcn_rsqf(fmaxf(__builtin_fmaf(-mean[i], mean[i], ex2), 0.f) + 1e-5f);
        }
    };
    // (x - mean) rstd gamma + beta for this lane's columns; which: 0 = (g1, b1), 1 = (g2, b2).  IN_PLACE: the stream itself is normalised
    // (chained LayerNorms); otherwise the result goes to the operand image (and the stream stays).
    auto normalise = [&](int which, const float (&mean)[MT], const float (&rstd)[MT], auto IN_PLACE) {
        constexpr bool in_place = decltype(IN_PLACE)::value;
        if constexpr (COCR_RC_EXP & 8192) return;
        const float *ga = lnp + which * 2 * D, *be = ga + D;
#pragma unroll
        for (int c = 0; c < NJ; ++c) {
            const int col = 256 * (c >> 1) + 32 * wave + 16 * (c & 1) + 4 * g;
            const f32x4 gv = *reinterpret_cast<const f32x4 *>(ga + col), bv = *reinterpret_cast<const f32x4 *>(be + col);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const float r = rstd[i], mr = -mean[i] * r;
                const f32x4 y = __builtin_elementwise_fma(__builtin_elementwise_fma(xs[i][c], (f32x4){r, r, r, r}, (f32x4){mr, mr, mr, mr}), gv, bv);
                if constexpr (in_place) {
                    xs[i][c] = y;
                } else {
                    const int row = 16 * i + r16;
                    const bf16x4 o = {(T)y[0], (T)y[1], (T)y[2], (T)y[3]};
                    *reinterpret_cast<bf16x4 *>(xa + (col >> 6) * PANEL + row * 128 + ((((col & 63) >> 3) ^ (row & 7)) << 4) + (col & 4) * 2) = o;
                }
            }
        }
    };
    // Residual + LayerNorm(s) of a stage whose product has left x_new in the stream registers.
    //   single:  x <- x_new ;            operand <- LN1(x)
    //   chained: x <- LN1(x_new) ;       operand <- LN2(x)          (block-final LayerNorm + the next block's first)
    // `pbuf`: 2 x BMC x PROW bytes of LDS nobody reads or writes during the epilogue (a hidden-chunk image that is not the last one read).
    auto rowln_epilogue = [&](const ChainStage &st, unsigned char *pbuf) {
        const bool chained = st.g2 != nullptr;
        if constexpr (TAPS) { if (st.tap_pre) store_stream(st.tap_pre); }
        float mean[MT], rstd[MT];
        if (!chained) {
            if (st.store_x) store_stream(p.x);
            row_stats(pbuf, true, mean, rstd);               // (barrier inside: every wave is also done reading the old operand image)
            normalise(0, mean, rstd, std::false_type{});
        } else {
            row_stats(pbuf, true, mean, rstd);
            normalise(0, mean, rstd, std::true_type{});
            if (st.store_x) store_stream(p.x);
            if constexpr (TAPS) { if (st.tap_post) store_stream(st.tap_post); }
            row_stats(pbuf + BMC * PROW, false, mean, rstd);
            normalise(1, mean, rstd, std::false_type{});
        }
        lds_fence_barrier();                                 // new operand image complete
        if (st.store_xn) {                                   // bf16 operand rows -> global (the decoder's input), coalesced from the image
            for (int id = tid; id < BMC * (D / 8); id += 512) {
                const int row = id / (D / 8), ch = id - row * (D / 8);
                if (m0 + row < mend)
                    copy16(p.xn + (size_t)(m0 + row) * D + ch * 8, reinterpret_cast<const T *>(xa + (ch >> 3) * PANEL + row * 128 + (((ch & 7) ^ (row & 7)) << 4)));
            }
        }
    };

    auto run_stage = [&](auto KIND, auto FIRST, const ChainStage &st, const T *after) {      // `after`: this wave's first slice of the next stage
        constexpr int kind = decltype(KIND)::value;
        if constexpr (kind == ST_ROWLN) {
            // x_new = x + bias + A W^T: NS column steps x KS k-slices
            request_ln_params(st);
            auto slice = [&](int ns, int ks) { return st.W + ((size_t)(ns * 8 + wave) * KS + ks) * SLICE; };
            if constexpr (decltype(FIRST)::value) {
                // first stage of a launch: the stream is still on its way -- the product starts from zero and the stream (+ bias) is
                // added afterwards
                f32x4 acc[MT][NJ];
#pragma unroll
                for (int ns = 0; ns < NS; ++ns)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const T *nxt = ks + 1 < KS ? slice(ns, ks + 1) : (ns + 1 < NS ? slice(ns + 1, 0) : after);
                        if (ks == 0) step(xa, acc, 2 * ns, nxt, no_side, std::true_type{}, KKR{});
                        else step(xa + ks * 4 * PANEL, acc, 2 * ns, nxt, no_side, std::false_type{}, KKR{});
                    }
                add_bias_to_stream(st.bias, 1.0f);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int c = 0; c < NJ; ++c) xs[i][c] += acc[i][c];
            } else {
                // accumulate straight onto the stream registers (x + bias = the product's initial accumulator)
                add_bias_to_stream(st.bias, 1.0f);
#pragma unroll
                for (int ns = 0; ns < NS; ++ns)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
                        step(xa + ks * 4 * PANEL, xs, 2 * ns, ks + 1 < KS ? slice(ns, ks + 1) : (ns + 1 < NS ? slice(ns + 1, 0) : after), no_side, std::false_type{}, KKR{});
            }
            RSTAMP()                               // product done
            rowln_epilogue(st, hs);
            RSTAMP()                               // epilogue done
        } else if constexpr (kind == ST_FRONT) {
            // x = bias + Z Wout^T with K = F C (the frontend's output linear, convolution.py:224,235-236; the flatten of (f, c) is a view of the
            // channel-last frontend output, the weight's columns were permuted at load), then the first block's LayerNorm.  24 slices of 256 k at
            // F C = 6144: slice ks + 1 is in flight (LDS-DMA into the other hidden image) while slice ks is multiplied; one barrier per slice.
            // The split-K GEMM + reduction this replaces ran 300 workgroups on all 256 CUs for 50 us; this form streams every weight byte once
            // per row block, like every other stage.
            static_assert(decltype(FIRST)::value, "FRONT is a first stage");
            request_ln_params(st);
            const int nsl = st.K / 256;
            auto slice = [&](int ns, int ks) { return st.W + ((size_t)(ns * 8 + wave) * nsl + ks) * SLICE; };
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int c = 0; c < NJ; ++c) xs[i][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            add_bias_to_stream(st.bias, 1.0f);
#pragma unroll 1
            for (int ks = 0; ks < nsl; ++ks) {
                // here: slice ks has landed and is published, every wave is done with the other image
                if (ks + 1 < nsl) front_dma(ks + 1);
                const unsigned char *img = hs + (ks & 1) * IMGH;
#pragma unroll
                for (int ns = 0; ns < NS; ++ns)
                    step(img, xs, 2 * ns, ns + 1 < NS ? slice(ns + 1, ks) : (ks + 1 < nsl ? slice(0, ks + 1) : after), no_side, std::false_type{}, KK8{});
                if (ks + 1 < nsl) {
                    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(RING) : "memory");      // the DMAs of slice ks + 1 are older than the last step's 16 ring loads
                    lds_fence_barrier();
                }
            }
            RSTAMP()                               // product done
            rowln_epilogue(st, hs + (nsl & 1) * IMGH);                     // partials in the image the last slice did not use
            RSTAMP()                               // epilogue done
        } else if constexpr (kind == ST_FFN) {
            // Software pipeline over the 256-wide hidden chunks:   P1(c): hidden(c) = xa W1(c)^T                  (KS steps)
            //                                                      P2(c-1): stream += silu(hidden(c-1)) W2(c-1)^T  (NS steps)  beside
            //                                                      S(c):  bias + SiLU of hidden(c) -> LDS          (VALU, folded into P2(c-1)'s k-steps)
            // weight stream order: W1(0), W1(1), W2(0), W1(2), W2(1), ..., W2(last).
            const int FF = st.N, nchunks = FF / 256;
            auto w1 = [&](int c, int ks) { return st.W + ((size_t)(c * 8 + wave) * KS + ks) * SLICE; };
            auto w2 = [&](int c, int ns) { return st.W2 + ((size_t)(ns * 8 + wave) * (FF / 32) + c * 8) * 1024; };
            f32x4 acc1[MT][2];
            Bias2 bb;
            auto silu_tile = [&](int tIdx, unsigned char *hb, auto PIN) {  // tile t = (row tile t / 2, column tile t % 2) of acc1 -> hb
                if constexpr (COCR_RC_EXP & 4096) return;
                const int i = tIdx >> 1, j = tIdx & 1;
                // PIN: called from a k-step -- the arithmetic below is pure, and instruction selection otherwise gathers all twelve tiles'
                // worth of it in one block ahead of the product's first k-step (seen in the ISA: only the LDS stores stayed inside the
                // steps), where nothing overlaps it; an empty volatile asm on the inputs keeps each tile between its step's fences
                if constexpr (decltype(PIN)::value) asm volatile("" : "+v"(acc1[i][j]));
                const int jj = 32 * wave + 16 * j + 4 * g;               // hidden column inside the chunk
                const int ch16 = (jj & 63) >> 3, row = 16 * i + r16;
                // packed fp32 arithmetic around the two transcendentals: 5 packed + 4 transcendental + 1 convert per value pair
                typedef float f32x2_t __attribute__((ext_vector_type(2)));
                const f32x4 v4 = acc1[i][j] + bb.v[j];
                unsigned packed[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x2_t v = {v4[2 * h], v4[2 * h + 1]};
                    f32x2_t e = v * (f32x2_t){-1.44269504088896340736f, -1.44269504088896340736f};
                    if constexpr (!(COCR_RC_EXP & 16)) e = (f32x2_t){__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])} + (f32x2_t){1.0f, 1.0f};
                    const f32x2_t o = (COCR_RC_EXP & 16) ? v * e : v * (f32x2_t){__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
                    packed[h] = __builtin_bit_cast(unsigned, __builtin_convertvector(o, bf16x2));
                }
                typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                *reinterpret_cast<u32x2_t *>(hb + (jj >> 6) * PANEL + row * 128 + ((ch16 ^ (row & 7)) << 4) + ((jj & 7) >> 2) * 8) = (u32x2_t){packed[0], packed[1]};
            };
            auto p1 = [&](int c, const T *then) {                        // hidden(c) -> acc1; `then`: the slice that follows W1(c)
                // Waves 4..7 (the SIMD partners of 0..3) run this matrix-only product at raised priority: they finish it first and do the
                // vector-heavy second product + SiLU while their partners are still in this one, instead of both waves of a SIMD being
                // in the same kind of phase at the same time (without it the older wave always wins the matrix pipe: 3.0 k / 4.4 k cycles
                // for this product, the partners then alone in a 2.4 k tail of the next; chunk iteration 10.7 k -> 9.9 k cycles)
                if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (ks == 0) step(xa, acc1, 0, KS > 1 ? w1(c, 1) : then, no_side, std::true_type{}, KKP1{});
                    else step(xa + ks * 4 * PANEL, acc1, 0, ks + 1 < KS ? w1(c, ks + 1) : then, no_side, std::false_type{}, KKP1{});
                }
                if (wave >= 4) __builtin_amdgcn_s_setprio(0);
            };
            request_ln_params(st);
            bb = load_bias(st.bias);
            p1(0, nchunks > 1 ? w1(1, 0) : w2(0, 0));                     // P1(0)
            RSTAMP()                                                      // P1(0) done
#pragma unroll
            for (int t2 = 0; t2 < 2 * MT; ++t2) silu_tile(t2, hs, std::false_type{});       // S(0)
            add_bias_to_stream(st.bias2, st.alpha);                       // stream + alpha b2 = the second product's initial accumulator (after P1(0):
                                                                          // as a launch's first stage the stream loads are still in flight before)
            RSTAMP()                                                      // S(0) done
            lds_fence_barrier();
            RSTAMP()                                                      // barrier
            for (int c = 1; c < nchunks; ++c) {
                bb = load_bias(st.bias + c * 256);
                p1(c, w2(c - 1, 0));                                      // P1(c)
                RSTAMP()                                                  // P1(c) done
                unsigned char *hb = hs + (c & 1) * IMGH;
#pragma unroll
                for (int ns = 0; ns < NS; ++ns)                           // P2(c-1) beside S(c): the 2 MT tiles spread over the NS x 8 k-steps
                    step(hs + ((c - 1) & 1) * IMGH, xs, 2 * ns, ns + 1 < NS ? w2(c - 1, ns + 1) : (c + 1 < nchunks ? w1(c + 1, 0) : w2(c, 0)),
                         [&](int kk) {
#pragma unroll
                             for (int t2 = 0; t2 < 2 * MT; ++t2) if ((t2 * 8 * NS) / (2 * MT) == ns * 8 + kk) silu_tile(t2, hb, std::true_type{});
                         }, std::false_type{}, KK8{});
                RSTAMP()                                                  // P2(c-1) + S(c) done
                lds_fence_barrier();
                RSTAMP()                                                  // barrier
            }
#pragma unroll
            for (int ns = 0; ns < NS; ++ns)                               // P2(last)
                step(hs + ((nchunks - 1) & 1) * IMGH, xs, 2 * ns, ns + 1 < NS ? w2(nchunks - 1, ns + 1) : after, no_side, std::false_type{}, KKL{});
            RSTAMP()                                                      // P2(last) done
            rowln_epilogue(st, hs + (nchunks & 1) * IMGH);               // partials in the hidden image the last product did not read
            RSTAMP()                                                      // epilogue done
        } else if constexpr (kind == ST_GLU) {
            // 2 D packed columns in steps of 256: wave w's pair = (value tile, gate tile) of channels 128 step + 16 w .. +15
            EpiGLU<T> e{st.out, D, st.bias, 2 * D};
            auto slice = [&](int s2, int ks) { return st.W + ((size_t)(s2 * 8 + wave) * KS + ks) * SLICE; };
            constexpr int NSTEP = 2 * D / 256;
#pragma unroll 1
            for (int s2 = 0; s2 < NSTEP; ++s2) {
                const Bias2 bb = load_bias(st.bias + s2 * 256);          // v[0]: value bias, v[1]: gate bias
                f32x4 acc[MT][2];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const T *nxt = ks + 1 < KS ? slice(s2, ks + 1) : (s2 + 1 < NSTEP ? slice(s2 + 1, 0) : after);
                    if (ks == 0) step(xa, acc, 0, nxt, no_side, std::true_type{}, KKG{});
                    else step(xa + ks * 4 * PANEL, acc, 0, nxt, no_side, std::false_type{}, KKG{});
                }
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int row = 16 * i + r16;
                    bf16x4 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) o[q] = (T)((acc[i][0][q] + bb.v[0][q]) * sigmoid_f(acc[i][1][q] + bb.v[1][q]));
                    *reinterpret_cast<bf16x4 *>(hs + row * OSD + (s2 * 128 + 16 * wave + 4 * g) * 2) = o;
                }
            }
            lds_fence_barrier();
            for (int id = tid; id < BMC * (D / 8); id += 512) {          // coalesced copy of the staged [BMC][D] tile
                const int row = id / (D / 8), ch = id - row * (D / 8);
                if (m0 + row < mend) e.store(m0 + row, ch * 8, reinterpret_cast<const T *>(hs + row * OSD + ch * 16), 8);
            }
        } else if constexpr (kind == ST_QKV) {   // 3 D columns in steps of 256; step s3 lies inside one of q, k, v
            auto slice = [&](int s3, int ks) { return st.W + ((size_t)(s3 * 8 + wave) * KS + ks) * SLICE; };
            constexpr int NSTEP = 3 * D / 256;
#pragma unroll
            for (int s3 = 0; s3 < NSTEP; ++s3) {
                const Bias2 bb = load_bias(st.bias + s3 * 256);
                f32x4 acc[MT][2];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const T *nxt = ks + 1 < KS ? slice(s3, ks + 1) : (s3 + 1 < NSTEP ? slice(s3 + 1, 0) : after);
                    if (ks == 0) step(xa, acc, 0, nxt, no_side, std::true_type{}, KKQ{});
                    else step(xa + ks * 4 * PANEL, acc, 0, nxt, no_side, std::false_type{}, KKQ{});
                }
                unsigned char *tile = hs + (s3 & 1) * (BMC * OS);
                if constexpr (COCR_RC_EXP & 1024) { asm volatile("" :: "v"(acc[0][0]), "v"(acc[MT - 1][1])); continue; }
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const int row = 16 * i + r16;
                        bf16x4 o;
#pragma unroll
                        for (int q = 0; q < 4; ++q) o[q] = (T)(acc[i][j][q] + bb.v[j][q]);
                        *reinterpret_cast<bf16x4 *>(tile + row * OS + (32 * wave + 16 * j + 4 * g) * 2) = o;
                    }
                RSTAMP()                                     // product + staging of step s3 done
                lds_fence_barrier();                         // tile s3 complete (tile s3-1 was flushed before this barrier)
                {   // thread -> one 8-column chunk of BMC / 16 rows
                    const int which = s3 / NS;
                    T *base = which == 0 ? st.q : (which == 1 ? st.k : st.v);
                    const int ch = tid & 31, hd = (s3 - which * NS) * 256 + ch * 8, hh = hd / p.dh, d = hd - hh * p.dh;
                    base += (size_t)hh * p.Tp * p.dhp + d;
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const int row = (tid >> 5) + 16 * i;
                        if (m0 + row < mend) copy16(base + rowoff[row], reinterpret_cast<const T *>(tile + row * OS + ch * 16));
                    }
                }
            }
        }
    };
    auto first_slice = [&](int i) -> const T * { return p.st[i].W + (size_t)wave * ((i == 0 && FRONT0) ? p.st[0].K / 256 : KS) * SLICE; };
    run_stage(std::integral_constant<int, K0>{}, std::true_type{}, p.st[0], K1 >= 0 ? first_slice(1) : first_slice(0));
    run_stage(std::integral_constant<int, K1>{}, std::false_type{}, p.st[1], K2 >= 0 ? first_slice(2) : first_slice(0));
    run_stage(std::integral_constant<int, K2>{}, std::false_type{}, p.st[2], K3 >= 0 ? first_slice(3) : first_slice(0));
    run_stage(std::integral_constant<int, K3>{}, std::false_type{}, p.st[3], first_slice(0));
    RSTAMP()                                       // end
#undef RSTAMP
}

template <int D, int MT, int DWK, int K0, int K1, int K2, int K3, bool TAPS, int KD = 8, int KL = 8>
static inline hipError_t launch_rowchain_mt(hipStream_t s, const ChainArgs &a) {
    constexpr size_t lds = rowchain_lds_bytes<D, MT>() + (rowchain_taps_place<D, MT, DWK>() == 2 ? (size_t)(DWK + 1) * 1024 : 0);
    static_assert(lds <= 160 * 1024, "LDS of one workgroup");
    auto kern = rowchain_kernel<D, MT, DWK, K0, K1, K2, K3, TAPS, KD, KL>;
    hipError_t e = raise_lds_limit((const void *)kern, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(ceil_div(a.M, 16 * MT)), dim3(512), lds, s, a);
    return hipGetLastError();
}

// Rows per workgroup.  Throughput (several batches in flight, `rows` = 0): the most rows the registers and LDS allow (the weight stream
// is read once per workgroup).  One batch in flight: the largest block that still gives every CU of the chip a workgroup.
template <int D> static inline int rowchain_pick_mt(int M, int rows_hint) {
    constexpr int MAXMT = D == 256 ? 6 : 4;
    if (rows_hint > 0) {
        const int mt = std::min(MAXMT, std::max(2, rows_hint / 16));
        return (D == 256 && (mt == 4 || mt == 5)) ? 3 : (D == 512 && mt == 3) ? 4 : mt;      // instantiated: 6, 3, 2 (D = 256); 4, 2 (D = 512)
    }
    if (M >= 16 * MAXMT * 50) return MAXMT;
    return 2;
}

// HAS_TAPS: whether the debug instantiation exists for this chain shape (the shapes the default forward uses)
template <int D, int DWK, int K0, int K1, int K2, int K3, bool HAS_TAPS>
static inline hipError_t launch_rowchain_cfg(hipStream_t s, const ChainArgs &a, bool taps, int rows_hint) {
    const int mt = rowchain_pick_mt<D>(a.M, rows_hint);
    if (taps && !HAS_TAPS) return hipErrorInvalidValue;
    // the reference's default model in its zero-padded 256 / 768-wide form: 5 of 8 k-steps per K = D product, 2 of 8 in the FFN's last chunk
    // (no debug instantiation: taps run the full-depth kernels, which compute the same values -- the skipped products are exact zeros)
    const bool narrow = D == 256 && a.kd == 5 && a.kl == 2 && !taps;
#define COCR_RC(MTV)                                                                                            \
    if (mt == MTV) {                                                                                            \
        if constexpr (HAS_TAPS) { if (taps) return launch_rowchain_mt<D, MTV, DWK, K0, K1, K2, K3, true>(s, a); } \
        if constexpr (D == 256) { if (narrow) return launch_rowchain_mt<D, MTV, DWK, K0, K1, K2, K3, false, 5, 2>(s, a); } \
        return launch_rowchain_mt<D, MTV, DWK, K0, K1, K2, K3, false>(s, a);                                     \
    }
    if constexpr (D == 256) { COCR_RC(6) COCR_RC(3) } else { COCR_RC(4) }
    COCR_RC(2)
#undef COCR_RC
    return hipErrorInvalidValue;
}

// the chain shapes the forward uses; stage weights point at the fragment-major copies
template <int D>
static inline hipError_t launch_rowchain(hipStream_t s, const ChainArgs &a, bool taps, int rows_hint) {
    const int k0 = a.st[0].kind, k1 = a.nstages > 1 ? a.st[1].kind : -1, k2 = a.nstages > 2 ? a.st[2].kind : -1, k3 = a.nstages > 3 ? a.st[3].kind : -1;
    if (a.dw_in) {                       // depthwise-conv prologue (kernel 31): the chains that follow the conv module's GLU
        if (k0 == ST_ROWLN && k1 == ST_FFN && k2 == ST_FFN && k3 == ST_QKV) return launch_rowchain_cfg<D, 31, ST_ROWLN, ST_FFN, ST_FFN, ST_QKV, true>(s, a, taps, rows_hint);
        if (k0 == ST_ROWLN && k1 == ST_FFN && k2 == -1) return launch_rowchain_cfg<D, 31, ST_ROWLN, ST_FFN, -1, -1, true>(s, a, taps, rows_hint);
        return hipErrorInvalidValue;
    }
    if (k0 == ST_FRONT && k1 == ST_FFN && k2 == ST_QKV && k3 == -1) return launch_rowchain_cfg<D, 0, ST_FRONT, ST_FFN, ST_QKV, -1, true>(s, a, taps, rows_hint);
    if (k0 == ST_FFN && k1 == ST_QKV && k2 == -1) return launch_rowchain_cfg<D, 0, ST_FFN, ST_QKV, -1, -1, true>(s, a, taps, rows_hint);
    // one feed-forward module alone (operand tile + stream in, LayerNorm epilogue, nothing else): the measurement probe of cocr_api.hip
    // (COCR_FFN_PROBE=1: matrix-pipe counters of the FFN products by themselves, tools/ffn_probe.py)
    if (k0 == ST_FFN && k1 == -1) return launch_rowchain_cfg<D, 0, ST_FFN, -1, -1, -1, false>(s, a, false, rows_hint);
    if (k0 == ST_ROWLN && k1 == ST_GLU && k2 == -1) return launch_rowchain_cfg<D, 0, ST_ROWLN, ST_GLU, -1, -1, true>(s, a, taps, rows_hint);
    // (without the depthwise prologue: conv kernels other than 31 -- the stand-alone depthwise kernel runs before the launch --, COCR_NO_DW_FUSE A/B runs)
    if (k0 == ST_ROWLN && k1 == ST_FFN && k2 == ST_FFN && k3 == ST_QKV) return launch_rowchain_cfg<D, 0, ST_ROWLN, ST_FFN, ST_FFN, ST_QKV, true>(s, a, taps, rows_hint);
    if (k0 == ST_ROWLN && k1 == ST_FFN && k2 == -1) return launch_rowchain_cfg<D, 0, ST_ROWLN, ST_FFN, -1, -1, true>(s, a, taps, rows_hint);
    return hipErrorInvalidValue;
}
