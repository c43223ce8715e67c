// Frontend stages F1-F3 of a factor-4 Conv2dSubsampling in ONE kernel (bf16 mode, C = 256 channels, F = 24 output rows):
//     Conv2d(1,C,3,s2,p1) + ReLU  ->  depthwise Conv2d(C,C,3,s2,p1)  ->  pointwise Conv2d(C,C,1) + ReLU
// (reference conformer/convolution.py:192-205).  Neither the (B,C,W/2,H/2) tensor Z1 nor the depthwise output Z2 is
// ever written: a workgroup owns 4 output frames of one line = 4 x 24 = 96 rows of the (B*T*F, C) activation, builds
// them as the 96 x 256 bf16 operand image of the chain kernels (rowchain.hip.h) and runs the pointwise conv on it with the
// same register-streamed weight ring.  All three convolutions run on the matrix cores:
//   * conv.0: (pixels x 9) x (9 x C), K padded to 16 as k = 4 dt + df -- a lane's 4 k-values are 4 consecutive image rows
//     of one image column: two 4-byte LDS reads from the transposed bf16 line tile (no im2col buffer); channels on the MFMA
//     row side, so a lane holds 4 consecutive channels of one Z1 pixel = one 8-byte store into the Z1 tile;
//   * depthwise conv.2: per block of 16 channels a (positions x (10 taps x 16 ch)) x ((10 taps x 16 ch) x 16 ch) product
//     whose B operand is block-diagonal (w2[c][tap] on the diagonal, tap 9 = 0): 15/16 of the multiplies hit zeros, but
//     the matrix pipe is otherwise idle here and the A operand is simply 8 consecutive channels of one Z1 pixel
//     (one ds_read_b128 per lane and k-chunk) -- the VALU form costs 9 packed FMAs + 6 LDS reads + conversions per output pair;
//   * pointwise conv.3: one step of the chain kernels' product (fragment-major weights, 16-fragment register ring).
// Z1 is held 64 channels at a time ([9 columns][50 rows][64 ch] bf16 tile), 4 passes.  Arithmetic: bf16 operands (pixels,
// all weights, Z1, Z2), fp32 accumulation -- the fp32 mode keeps the VALU kernel + GEMM.
#pragma once
#include "gemm.hip.h"
#include "conv.hip.h"

// ---- weight images built once per model (cocr_api: ensure_packed) -----------------------------------------------
// w0f: conv.0 A-fragments [16 channel tiles][64 lanes][4 bf16]: lane (r, g) = w0[16 nt + r][dt = g][df = 0..2], 0
// dwf: depthwise B-fragments [16 blocks][5 k-chunks][64 lanes][8 bf16]: lane (r, g), element i <-> k = 32 chunk + 8 g + i
//      = (tap = 2 chunk + (g >> 1), c' = 8 (g & 1) + i); value w2[16 blk + r][tap] where c' == r and tap < 9, else 0
__global__ __launch_bounds__(256) void frontend_pack_kernel(const float *__restrict__ w0, const float *__restrict__ w2, bf16_t *__restrict__ w0f,
                                                            bf16_t *__restrict__ dwf, int C) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < (C / 16) * 64 * 4) {
        const int e = i & 3, lane = (i >> 2) & 63, nt = i >> 8, r = lane & 15, g = lane >> 4;
        w0f[i] = (bf16_t)((g < 3 && e < 3) ? w0[(16 * nt + r) * 9 + 3 * g + e] : 0.0f);
    }
    if (i < (C / 16) * 5 * 64 * 8) {
        const int e = i & 7, lane = (i >> 3) & 63, chunk = (i >> 9) % 5, blk = i / (5 * 512), r = lane & 15, g = lane >> 4;
        const int tap = 2 * chunk + (g >> 1), cc = 8 * (g & 1) + e;
        dwf[i] = (bf16_t)((cc == r && tap < 9) ? w2[(16 * blk + r) * 9 + tap] : 0.0f);
    }
}

// ReLU of four bf16 values as two packed signed-16-bit maxima with 0: a bf16 is negative iff its bit pattern is a negative int16, and
// rounding commutes with ReLU (round-to-nearest keeps the sign), so this equals rounding max(x, 0).  fmaxf / fmed3 on the fp32 values
// cost 8 VALU operations per 4 values here (the compiler quiets a possible signalling NaN in front of each maximum).
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned relu_pack2(float a, float b) {        // bf16(max(a, 0)) | bf16(max(b, 0)) << 16
    typedef short s16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const bf16x2 h = __builtin_convertvector((f32x2_t){a, b}, bf16x2);   // one v_cvt_pk_bf16_f32
    const s16x2_t zero = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, h), zero));
}
__device__ __forceinline__ u32x2_t relu_bf16x4(const f32x4 &v) { return (u32x2_t){relu_pack2(v[0], v[1]), relu_pack2(v[2], v[3])}; }

template <typename TIn>
__global__ __launch_bounds__(512) void frontend96_kernel(const TIn *__restrict__ X, int H, int W, int T1, int F1, int Tn,
                                                         const bf16_t *__restrict__ w0f, const float *__restrict__ b0,
                                                         const bf16_t *__restrict__ dwf, const float *__restrict__ b2,
                                                         const bf16_t *__restrict__ wpw,        // fragment-major (256, 256)
                                                         const float *__restrict__ bpw, bf16_t *__restrict__ Z3, int HS, int N, unsigned long long *stamps) {
    typedef bf16_t T;
    constexpr int C = 256, F = 24, TB = 4, BMC = TB * F, MT = 6, KC1 = C / 32;
    constexpr int NA = 2 * TB + 1, NCOL = 4 * TB + 4, ZR = 2 * F + 2, ZC = 72;      // Z1 tile: rows zr = f1 + 1 in [0, 2F + 1], 64 channels + pad
    constexpr int PANEL = BMC * 128, IMG = 4 * PANEL, OS = C * 2 + 16, SLICE = 16 * 512;
    constexpr int XAB = BMC * OS > IMG ? BMC * OS : IMG;                  // the operand image's area also takes the bf16 output tile [96][OS]
    static_assert(BMC == 96, "operand image of the chain kernels");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *xa = smem;                                             // 96 x 256 bf16 operand image (Z2)
    T *z1 = reinterpret_cast<T *>(smem + XAB);                            // [NA][ZR][ZC]
    T *xs = z1 + (NA * ZR + 1) * ZC;                                      // (one dump row behind the Z1 tile) [NCOL][HS] transposed line tile: row rr <-> image row rr - 1
    unsigned char *prm = reinterpret_cast<unsigned char *>(xs + NCOL * HS);   // conv.0 A-fragments (8 KiB), b0 (1 KiB), b2 (1 KiB); 16-byte aligned (HS % 4 == 0)
    const T *w0s = reinterpret_cast<const T *>(prm);
    const float *b0s = reinterpret_cast<const float *>(prm + 8192), *b2s = b0s + 256;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4, swz = r16 & 7;
    // PERSISTENT: one workgroup per CU walks the (line, 4-frame block) items blockIdx.x, blockIdx.x + gridDim.x, ...  With one workgroup
    // per CU nothing else hides an item's input latency (a 20-column strip of the line: 96 short row segments from HBM, ~3 us of the
    // ~14 us an item took when every item was a workgroup of its own), so the NEXT item's pixels are requested during the last
    // channel pass of the current one and wait in 4 registers per thread; the parameters go to LDS once.
    const int nblk = (Tn + TB - 1) / TB, nitems = nblk * N, stride = gridDim.x;
    int item = blockIdx.x;
#ifdef COCR_CHAIN_STAMPS_BUILD
    int nstamp = 0;
    auto stamp = [&]() {
        if (stamps && blockIdx.x == 1 && tid == 0 && nstamp < 40) stamps[nstamp] = __builtin_readcyclecounter();
        ++nstamp;
    };
#else
    auto stamp = [&]() {};
#endif
    stamp();

    // ---- line tile (transposed, bf16, zero outside the image): lanes run along the image row (coalesced), all of a thread's
    // loads are issued before the first use
    constexpr int FILL = 4;                                 // NCOL * HS <= 4 * 512 (HS <= 100: checked by the launcher)
    TIn fv[FILL];                                           // raw pixels (converted and masked when they are written to LDS: no use, no wait, here)
    auto load_pixels = [&](int it) {
        const int bb = it / nblk, c0 = 4 * (it - bb * nblk) * TB - 3;
        const TIn *Xb = X + (size_t)bb * H * W;
        int tv = tid;
        asm volatile("" : "+v"(tv));                        // (index arithmetic recomputed per item: hoisted out of the item loop it costs registers the pointwise stage needs)
#pragma unroll
        for (int u = 0; u < FILL; ++u) {
            const int i = tv + 512 * u, rr = i / NCOL, ci = i - rr * NCOL;
            const int w = min(max(c0 + ci, 0), W - 1), r = min(max(rr - 1, 0), H - 1);      // clamped address; the select is made at the write
            fv[u] = Xb[(size_t)r * W + w];
        }
    };
    if (item < nitems) load_pixels(item);
    // ---- conv.0 fragments and the two bias vectors -> LDS (10 wave-instructions of 1 KiB), once
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(w0f + wave * 512 + lane * 8), (lds_ptr_t)(prm + wave * 1024), 16, 0, 0);
    if (wave < 2) __builtin_amdgcn_global_load_lds((gbl_ptr_t)((wave ? b2 : b0) + lane * 4), (lds_ptr_t)(prm + 8192 + wave * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int FT = (F1 + 15) >> 4, ntile = NA * FT;        // 16-pixel tiles of the Z1 columns
    // depthwise work of this wave in every pass: channel block blk of the pass, position tiles mq, mq + 2, mq + 4
    const int blk = wave & 3, mq = wave >> 2;

    for (; item < nitems; item += stride) {
    const int b = item / nblk, t0 = (item - b * nblk) * TB;
    int tv = tid;
    asm volatile("" : "+v"(tv));
    bf16x8 ring[16];                                       // pointwise-conv weights of this wave's 32 output channels: requested in the last pass
    // the Z1 tile's zero rows (the depthwise conv's padding)
    for (int i = tv; i < NA * ZC; i += 512) {
        const int a = i / ZC, c = i - a * ZC;
        z1[(a * ZR) * ZC + c] = (T)0.0f;
        for (int zr = F1 + 1; zr < ZR; ++zr) z1[(a * ZR + zr) * ZC + c] = (T)0.0f;
    }
#pragma unroll
    for (int u = 0; u < FILL; ++u) {
        const int i = tv + 512 * u, rr = i / NCOL, ci = i - rr * NCOL, c = 4 * t0 - 3 + ci;
        const bool inside = c >= 0 && c < W && rr >= 1 && rr <= H;
        if (i < NCOL * HS) xs[ci * HS + rr] = (T)(inside ? pixel_to_f32<TIn>(fv[u]) : 0.0f);
    }
    stamp();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    stamp();

    // conv.0 work of this wave in every pass: pixel tiles wave, wave + 8, ... (at most 4) x the pass's 4 channel tiles; the pixel
    // fragments do not depend on the pass
    s16x4 pf[4];
    bool valid[4];
    int zoff[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int mt = min(wave + 8 * u, ntile - 1);
        const int a = mt / FT, f1 = 16 * (mt - a * FT) + r16;            // this lane's pixel: Z1 column a, row f1
        const int t1 = 2 * t0 - 1 + a;
        const T *px = xs + (2 * a + g) * HS + 2 * f1;                    // image rows 2 f1 - 1 .. 2 f1 + 2 of image column col0 + 2a + dt
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        pf[u] = __builtin_bit_cast(s16x4, (u32x2){*reinterpret_cast<const unsigned *>(px), *reinterpret_cast<const unsigned *>(px + 2)});
        valid[u] = t1 >= 0 && t1 < T1 && f1 < F1;
        zoff[u] = (wave + 8 * u < ntile && f1 + 1 < ZR) ? (a * ZR + f1 + 1) * ZC : NA * ZR * ZC;      // else: the dump row behind the tile
    }
    const bool all_valid = __builtin_amdgcn_ballot_w64(valid[0] && valid[1] && valid[2] && valid[3]) == ~0ull;
    // depthwise fragments of (pass, blk): one pass ahead in registers
    bf16x8 dwn[5];
    auto request_dw = [&](int pass) {
#pragma unroll
        for (int ch = 0; ch < 5; ++ch) dwn[ch] = *reinterpret_cast<const bf16x8 *>(dwf + ((size_t)(4 * pass + blk) * 5 + ch) * 512 + lane * 8);
    };
    request_dw(0);
    auto do_pass = [&](int pass, auto LAST) {
        const int cb = 64 * pass;
        bf16x8 dwb[5];
#pragma unroll
        for (int ch = 0; ch < 5; ++ch) dwb[ch] = dwn[ch];
        if (pass < 3) request_dw(pass + 1);
        const f32x4 bias2 = *reinterpret_cast<const f32x4 *>(b2s + cb + 16 * blk + 4 * g);
        s16x4 wf[4];
        f32x4 bias0[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            wf[nt] = *reinterpret_cast<const s16x4 *>(w0s + ((size_t)((cb >> 4) + nt) * 64 + lane) * 4);
            bias0[nt] = *reinterpret_cast<const f32x4 *>(b0s + cb + 16 * nt + 4 * g);
        }
        // ---- conv.0 + ReLU -> Z1 tile
        f32x4 c0[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) c0[u][nt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wf[nt], pf[u], bias0[nt], 0, 0, 0);
        // (interior workgroups -- all pixels of all four tiles inside Z1 -- skip the per-value zero select: one uniform branch per
        // pass around two straight-line copies of the epilogue)
        auto z1_epilogue = [&](auto MASKED) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    u32x2 ob = relu_bf16x4(c0[u][nt]);
                    if (decltype(MASKED)::value && !valid[u]) ob = (u32x2){0u, 0u};      // outside Z1: the depthwise conv's zero padding
                    *reinterpret_cast<u32x2 *>(z1 + zoff[u] + 16 * nt + 4 * g) = ob;       // (tiles beyond the last: a dump row)
                }
        };
        if (all_valid) z1_epilogue(std::false_type{}); else z1_epilogue(std::true_type{});
        if constexpr (decltype(LAST)::value) {
            // the conv.0 accumulators are dead: the pointwise weights (used after this pass) and, behind them (loads return in order), the
            // next item's pixels
            const T *first = wpw + (size_t)wave * SLICE;
#pragma unroll
            for (int f = 0; f < 16; ++f) ring[f] = *reinterpret_cast<const bf16x8 *>(first + f * 512 + lane * 8);
            if (item + stride < nitems) load_pixels(item + stride);
        }
        stamp();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stamp();
        // ---- depthwise conv.2 of channels cb + 16 blk .. +15 for position tiles mq, mq + 2, mq + 4 -> operand image
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int p = 16 * (mq + 2 * u) + r16, tl = p / F, f = p - tl * F;    // output position (frame tl, row f)
            f32x4 acc = bias2;
#pragma unroll
            for (int ch = 0; ch < 5; ++ch) {
                const int tap = min(2 * ch + (g >> 1), 8), dt = tap / 3, df = tap - 3 * dt;
                const bf16x8 za = *reinterpret_cast<const bf16x8 *>(z1 + ((2 * tl + dt) * ZR + 2 * f + df) * ZC + 16 * blk + 8 * (g & 1));
                acc = mma16(dwb[ch], za, acc);
            }
            const bf16x4 o = {(T)acc[0], (T)acc[1], (T)acc[2], (T)acc[3]};
            const int col = cb + 16 * blk + 4 * g;
            *reinterpret_cast<bf16x4 *>(xa + (col >> 6) * PANEL + p * 128 + ((((col & 63) >> 3) ^ (p & 7)) << 4) + ((col & 7) >> 2) * 8) = o;
        }
        stamp();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // Z1 tile free for the next pass; after the last pass the operand image is complete
    };
#pragma unroll 1
    for (int pass = 0; pass < 3; ++pass) do_pass(pass, std::false_type{});
    do_pass(3, std::true_type{});                          // (a copy of its own: the weight ring is live only from here on)

    // ---- pointwise conv.3 + ReLU: one step of the chain product, this wave's 32 output channels
    const f32x4 bp0 = *reinterpret_cast<const f32x4 *>(bpw + 32 * wave + 4 * g), bp1 = *reinterpret_cast<const f32x4 *>(bpw + 32 * wave + 16 + 4 * g);
    f32x4 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i) { acc[i][0] = bp0; acc[i][1] = bp1; }
#pragma unroll
    for (int kk = 0; kk < KC1; ++kk) {
        bf16x8 a[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = lds_frag_swz(xa + (kk >> 1) * PANEL + (16 * i + r16) * 128, kk & 1, g, swz, T());
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i][j] = mma16(ring[2 * kk + j], a[i], acc[i][j]);
    }
    stamp();
    // The output tile takes the operand image's place (every wave has read its last fragment of it: one barrier), NOT the Z1 tile's:
    // the next item's line tile, zero rows and conv.0 passes then start while this item's rows are still being copied out, and the
    // item needs no closing barrier (the image is next written by the depthwise stage of the next item, two barriers on).
    __builtin_amdgcn_s_barrier();
    unsigned char *tile = xa;                                             // bf16 [96][OS]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = 16 * i + r16;
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<u32x2 *>(tile + row * OS + (32 * wave + 16 * j + 4 * g) * 2) = relu_bf16x4(acc[i][j]);
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    stamp();
    // rows (frame tl, f) of this line are consecutive rows of the (B*T*F, C) output: coalesced 16-byte stores
    T *out = Z3 + ((size_t)b * Tn + t0) * F * C;
    for (int id = tv; id < BMC * 32; id += 512) {
        const int row = id >> 5, ch = id & 31;
        if (t0 + row / F < Tn) copy16(out + (size_t)row * C + ch * 8, reinterpret_cast<const T *>(tile + row * OS + ch * 16));
    }
    stamp();
    }
}

static inline bool frontend96_supported(int C, int F1, int F2, int H) { return C == 256 && F2 == 24 && F1 <= 2 * F2 && H <= 4 * F2; }

template <typename TIn>
static inline hipError_t launch_frontend96(hipStream_t s, const TIn *X, int N, int H, int W, int T1, int F1, int Tn, const bf16_t *w0f, const float *b0,
                                           const bf16_t *dwf, const float *b2, const bf16_t *wpw, const float *bpw, bf16_t *Z3, unsigned long long *stamps = nullptr) {
    const int FT = (F1 + 15) / 16, HS = 2 * 16 * FT + 4;
    const size_t z1b = (size_t)(9 * 50 + 1) * 72 * 2, tileb = (size_t)96 * (256 * 2 + 16);
    const size_t lds = std::max((size_t)4 * 96 * 128, tileb) + z1b + (size_t)20 * HS * 2 + 10240;
    auto kern = frontend96_kernel<TIn>;
    hipError_t e = raise_lds_limit((const void *)kern, lds + 4096);
    if (e != hipSuccess) return e;
    static int ncu = 0;                                    // one persistent workgroup per CU (the kernel's LDS allows no second one)
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
        ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int nitems = ((Tn + 3) / 4) * N;
    hipLaunchKernelGGL(kern, dim3(std::min(nitems, ncu)), dim3(512), lds, s, X, H, W, T1, F1, Tn, w0f, b0, dwf, b2, wpw, bpw, Z3, HS, N, stamps);
    return hipGetLastError();
}
