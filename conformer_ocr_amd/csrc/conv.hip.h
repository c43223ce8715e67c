// Convolution kernels of the path (all VALU, fp32 arithmetic):
//   * frontend_conv12_kernel: Conv2d(1,C,3,s2,p1)+ReLU fused with the depthwise Conv2d(C,C,3,s2,p1)
//     (reference conformer/convolution.py:192-205); the (B,C,W/2,H/2) intermediate is never written.
//   * dw3x3s2_kernel: the depthwise stage of further subsampling steps (subsampling_factor > 4).
//   * dwconv_bn_silu_kernel: the conv module's DepthwiseConv1d(k) + BatchNorm1d(eval) + SiLU
//     (convolution.py:140-142) with the BatchNorm folded into taps and bias at load time.
#pragma once
#include "common.hip.h"

// ---- frontend F1+F2 ---------------------------------------------------------------------------
// Geometry: X[b][r][w] image row r (height H), column w (width W).  The reference
// transposes the line to (W,H) before the 2-D convs, so kernel dim 0 runs along image width (time).
//   Z1[c][t1][f1] = relu(b0[c] + sum_{dt,df} w0[c][dt][df] X[2 f1 + df - 1][2 t1 + dt - 1])   t1 < T1, f1 < F1
//   Z2[c][t][f]   = b2[c] + sum_{dt,df} w2[c][dt][df] Z1[c][2t + dt - 1][2f + df - 1]          (zero outside)
// Output layout (B, T, F, C): channel fastest, so the pointwise conv that follows is a plain
// row-major GEMM over (B*T*F) rows and the flatten (b,t,(f,c)) feeding the output linear is a view.
//
// Work split: a workgroup takes TB consecutive t of one line; its 256 threads are (t_local, channel)
// pairs.  The 4*TB+3 image columns the group needs are staged once in LDS as fp32 (transposed,
// [column][row+1], zero borders), every thread walks f = 0..F-1 keeping the previous Z1 column.
template <typename T> struct Pair;
template <> struct Pair<bf16_t> { typedef bf16x2 type; };
template <> struct Pair<float> { typedef f32x2 type; };

template <typename TIn> __device__ __forceinline__ float pixel_to_f32(TIn v);
template <> __device__ __forceinline__ float pixel_to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float pixel_to_f32<uint8_t>(uint8_t v) { return (float)v / 255.0f; }

// ---- frontend of subsampling_factor 2: conv.0 + ReLU only (convolution.py:192-198; no depthwise / pointwise stage follows) --------
//   Z1[b][t1][f1][c] = relu(b0[c] + sum_{dt,df} w0[c][dt][df] X[b][2 f1 + df - 1][2 t1 + dt - 1])      channel-last, like Z3 elsewhere
// One thread per (position, channel pair); a workgroup's 256 threads cover 512 / C positions.  HBM-bound on its output
// (N T1 F1 C elements against N H W pixels read); not a measured configuration (no reference hyper-parameter set uses factor 2).
template <typename T, typename TIn>
__global__ __launch_bounds__(256) void frontend_conv0_kernel(const TIn *__restrict__ X, int H, int W, int T1, int F1, int C, size_t npos,
                                                             const float *__restrict__ w0, const float *__restrict__ b0, T *__restrict__ Z1) {
    const int C2 = C >> 1;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t pos = idx / C2;
    if (pos >= npos) return;
    const int c = 2 * (int)(idx - pos * C2);
    const int f1 = (int)(pos % F1), t1 = (int)((pos / F1) % T1);
    const size_t b = pos / ((size_t)F1 * T1);
    const TIn *Xb = X + b * H * W;
    f32x2 acc = {b0[c], b0[c + 1]};
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int df = 0; df < 3; ++df) {
            const int w = 2 * t1 + dt - 1, r = 2 * f1 + df - 1;
            const float v = (w >= 0 && w < W && r >= 0 && r < H) ? pixel_to_f32<TIn>(Xb[(size_t)min(max(r, 0), H - 1) * W + min(max(w, 0), W - 1)]) : 0.0f;
            acc[0] = fmaf(w0[c * 9 + 3 * dt + df], v, acc[0]);
            acc[1] = fmaf(w0[(c + 1) * 9 + 3 * dt + df], v, acc[1]);
        }
    typedef typename Pair<T>::type P2;
    *reinterpret_cast<P2 *>(Z1 + pos * C + c) = (P2){(T)fmaxf(acc[0], 0.f), (T)fmaxf(acc[1], 0.f)};
}

template <typename T, typename TIn>
__global__ __launch_bounds__(256) void frontend_conv12_kernel(const TIn *__restrict__ X, int H, int W, int T1, int F1, int Tn, int F, int C,
                                                              const float *__restrict__ w0, const float *__restrict__ b0,
                                                              const float *__restrict__ w2, const float *__restrict__ b2,
                                                              T *__restrict__ Z2, int TB) {
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [ncols][HS]: LDS row rr <-> image row rr - 1, zero borders
    const int b = blockIdx.y, t0 = blockIdx.x * TB;
    const int ncols = 4 * TB + 3, HS = (H + 11) & ~3;            // rows -1 .. H+6 at least, multiple of 4 (16-byte row groups)
    const int col0 = 4 * t0 - 3;
    const TIn *Xb = X + (size_t)b * H * W;
    for (int i = threadIdx.x; i < ncols * HS; i += blockDim.x) {
        const int ci = i / HS, rr = i - ci * HS;
        const int w = min(max(col0 + ci, 0), W - 1), r = min(max(rr - 1, 0), H - 1);      // clamped address + select: no branch around the load
        const float v = pixel_to_f32<TIn>(Xb[(size_t)r * W + w]);
        xs[i] = (col0 + ci >= 0 && col0 + ci < W && rr >= 1 && rr <= H) ? v : 0.0f;
    }
    __syncthreads();
    // Each thread owns TWO adjacent channels of one t: every multiply-add is a packed v_pk_fma_f32 (the kernel is VALU-issue
    // bound: 63 fused multiply-adds per output and channel), the pixel operand is shared by both halves.  C is even.
    const int C2 = C >> 1;
    for (int idx = threadIdx.x; idx < TB * C2; idx += blockDim.x) {
        const int tl = idx / C2, c = 2 * (idx - tl * C2);
        const int t = t0 + tl;
        if (t >= Tn) continue;
        f32x2 k0[9], k2[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) { k0[i] = (f32x2){w0[c * 9 + i], w0[(c + 1) * 9 + i]}; k2[i] = (f32x2){w2[c * 9 + i], w2[(c + 1) * 9 + i]}; }
        const f32x2 bias0 = {b0[c], b0[c + 1]}, bias2 = {b2[c], b2[c + 1]};
        // Image rows 4f-1 .. 4f+2 of the 7 columns this (t, .) needs are one aligned 16-byte LDS group per column (row group f);
        // output f uses row groups f and f+1: one ds_read_b128 per column per f (wave-uniform address: a broadcast read).
        const float *xc = xs + (4 * tl) * HS;                // column ci of this t: xc + ci * HS
        f32x4 cur[7], nxt[7];
#pragma unroll
        for (int ci = 0; ci < 7; ++ci) cur[ci] = *reinterpret_cast<const f32x4 *>(xc + ci * HS);
        // validity of the three Z1 columns (t1 = 2t-1+a) is per thread, of the Z1 rows (f1) per step
        const bool tv[3] = {2 * t - 1 >= 0 && 2 * t - 1 < T1, 2 * t < T1, 2 * t + 1 < T1};
        const f32x2 zero2 = {0.f, 0.f};
        f32x2 prev[3] = {zero2, zero2, zero2};                // Z1[.][f1 = 2f - 1], f = 0: outside
        typedef typename Pair<T>::type P2;
        T *out = Z2 + (((size_t)b * Tn + t) * F) * C + c;
        for (int f = 0; f < F; ++f) {
#pragma unroll
            for (int ci = 0; ci < 7; ++ci) nxt[ci] = *reinterpret_cast<const f32x4 *>(xc + ci * HS + 4 * (f + 1));
            f32x2 s = bias2;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                // Z1(a, 2f): image rows 4f-1+df = element df of the current group; Z1(a, 2f+1): rows 4f+1+df = elements 2, 3 and next[0]
                f32x2 m = bias0, n = bias0;
#pragma unroll
                for (int dt = 0; dt < 3; ++dt) {
                    const f32x4 x0 = cur[2 * a + dt];
                    const float x4 = nxt[2 * a + dt][0];
                    m += k0[dt * 3 + 0] * (f32x2){x0[0], x0[0]} + k0[dt * 3 + 1] * (f32x2){x0[1], x0[1]} + k0[dt * 3 + 2] * (f32x2){x0[2], x0[2]};
                    n += k0[dt * 3 + 0] * (f32x2){x0[2], x0[2]} + k0[dt * 3 + 1] * (f32x2){x0[3], x0[3]} + k0[dt * 3 + 2] * (f32x2){x4, x4};
                }
                const bool vm = tv[a] && 2 * f < F1, vn = tv[a] && 2 * f + 1 < F1;          // zero padding of the second conv outside Z1
                m = vm ? (f32x2){fmaxf(m[0], 0.0f), fmaxf(m[1], 0.0f)} : zero2;
                n = vn ? (f32x2){fmaxf(n[0], 0.0f), fmaxf(n[1], 0.0f)} : zero2;
                s += k2[a * 3 + 0] * prev[a] + k2[a * 3 + 1] * m + k2[a * 3 + 2] * n;
                prev[a] = n;
            }
            P2 o; o[0] = from_f32<T>(s[0]); o[1] = from_f32<T>(s[1]);
            *reinterpret_cast<P2 *>(out + (size_t)f * C) = o;
#pragma unroll
            for (int ci = 0; ci < 7; ++ci) cur[ci] = nxt[ci];
        }
    }
}

typedef unsigned u32x2_conv __attribute__((ext_vector_type(2)));
// bf16(max(v, 0)) of four values as two packed registers
__device__ __forceinline__ u32x2_conv relu_pack4(const f32x4 &v) {
    const bf16x2 lo = {(bf16_t)fmaxf(v[0], 0.0f), (bf16_t)fmaxf(v[1], 0.0f)}, hi = {(bf16_t)fmaxf(v[2], 0.0f), (bf16_t)fmaxf(v[3], 0.0f)};
    return (u32x2_conv){__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
}

// ---- frontend F1+F2+F3 for 32 conv channels (the reference's default model, default_specs.py:48-61), bf16 mode (round 4) --------------
// frontend_conv12_kernel's walk (16 frames x 16 channel pairs per workgroup) with the pointwise Conv2d(32,32,1) + ReLU
// (convolution.py:203-205) behind it in the same launch: the workgroup's Z2 rows (16 F positions x 32 channels, bf16 -- the rounding
// point the separate GEMM had) stay in LDS and are the B operands of one v_mfma_f32_16x16x32_bf16 per (16 positions, 16 output channels)
// -- K = 32 is a single k-chunk --, the Z3 rows leave through the same tile as whole 64-byte rows.  Replaces two whole-chip launches
// (35.6 us + a 23 us GEMM that read and wrote 14.7 MB each at 1.3 TB/s) by one; Z2 never exists in memory.
template <typename TIn>
__global__ __launch_bounds__(256) void frontend_conv12pw32_kernel(const TIn *__restrict__ X, int H, int W, int T1, int F1, int Tn, int F,
                                                                  const float *__restrict__ w0, const float *__restrict__ b0,
                                                                  const float *__restrict__ w2, const float *__restrict__ b2,
                                                                  const bf16_t *__restrict__ wpw, const float *__restrict__ bpw, bf16_t *__restrict__ Z3) {
    typedef bf16_t T;
    constexpr int C = 32, TB = 16, C2 = C / 2, ZP = 80;          // ZP: bytes per Z2 row in LDS (64 + 16: the fragment reads of 16 rows spread over the banks)
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [ncols][HS]: LDS row rr <-> image row rr - 1, zero borders
    const int b = blockIdx.y, t0 = blockIdx.x * TB;
    const int ncols = 4 * TB + 3, HS = (H + 11) & ~3;
    unsigned char *z2s = reinterpret_cast<unsigned char *>(xs + ncols * HS);      // [TB * F][ZP]
    const int col0 = 4 * t0 - 3;
    const TIn *Xb = X + (size_t)b * H * W;
    for (int i = threadIdx.x; i < ncols * HS; i += 256) {
        const int ci = i / HS, rr = i - ci * HS;
        const int w = min(max(col0 + ci, 0), W - 1), r = min(max(rr - 1, 0), H - 1);
        const float v = pixel_to_f32<TIn>(Xb[(size_t)r * W + w]);
        xs[i] = (col0 + ci >= 0 && col0 + ci < W && rr >= 1 && rr <= H) ? v : 0.0f;
    }
    __syncthreads();
    {
        const int tl = threadIdx.x / C2, c = 2 * (threadIdx.x - tl * C2);
        const int t = t0 + tl;
        f32x2 k0[9], k2[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) { k0[i] = (f32x2){w0[c * 9 + i], w0[(c + 1) * 9 + i]}; k2[i] = (f32x2){w2[c * 9 + i], w2[(c + 1) * 9 + i]}; }
        const f32x2 bias0 = {b0[c], b0[c + 1]}, bias2 = {b2[c], b2[c + 1]};
        const float *xc = xs + (4 * tl) * HS;
        f32x4 cur[7], nxt[7];
#pragma unroll
        for (int ci = 0; ci < 7; ++ci) cur[ci] = *reinterpret_cast<const f32x4 *>(xc + ci * HS);
        const bool tv[3] = {2 * t - 1 >= 0 && 2 * t - 1 < T1, 2 * t < T1, 2 * t + 1 < T1};
        const f32x2 zero2 = {0.f, 0.f};
        f32x2 prev[3] = {zero2, zero2, zero2};
        unsigned char *out = z2s + (size_t)(tl * F) * ZP + c * 2;
        for (int f = 0; f < F; ++f) {                          // (same arithmetic, in the same order, as frontend_conv12_kernel)
#pragma unroll
            for (int ci = 0; ci < 7; ++ci) nxt[ci] = *reinterpret_cast<const f32x4 *>(xc + ci * HS + 4 * (f + 1));
            f32x2 s = bias2;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                f32x2 m = bias0, n = bias0;
#pragma unroll
                for (int dt = 0; dt < 3; ++dt) {
                    const f32x4 x0 = cur[2 * a + dt];
                    const float x4 = nxt[2 * a + dt][0];
                    m += k0[dt * 3 + 0] * (f32x2){x0[0], x0[0]} + k0[dt * 3 + 1] * (f32x2){x0[1], x0[1]} + k0[dt * 3 + 2] * (f32x2){x0[2], x0[2]};
                    n += k0[dt * 3 + 0] * (f32x2){x0[2], x0[2]} + k0[dt * 3 + 1] * (f32x2){x0[3], x0[3]} + k0[dt * 3 + 2] * (f32x2){x4, x4};
                }
                const bool vm = tv[a] && 2 * f < F1, vn = tv[a] && 2 * f + 1 < F1;
                m = vm ? (f32x2){fmaxf(m[0], 0.0f), fmaxf(m[1], 0.0f)} : zero2;
                n = vn ? (f32x2){fmaxf(n[0], 0.0f), fmaxf(n[1], 0.0f)} : zero2;
                s += k2[a * 3 + 0] * prev[a] + k2[a * 3 + 1] * m + k2[a * 3 + 2] * n;
                prev[a] = n;
            }
            *reinterpret_cast<bf16x2 *>(out + (size_t)f * ZP) = (bf16x2){(T)s[0], (T)s[1]};      // (frames beyond T: finite values nobody stores)
#pragma unroll
            for (int ci = 0; ci < 7; ++ci) cur[ci] = nxt[ci];
        }
    }
    __syncthreads();
    // ---- pointwise conv + bias + ReLU: position tiles wave, wave + 4, ...; weights on the MFMA row side (a lane ends up with 4 consecutive
    // output channels of one position)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    const int ntile = TB * F / 16;
    const bf16x8 wf0 = load_frag(wpw + (size_t)r16 * C + 8 * g), wf1 = load_frag(wpw + (size_t)(16 + r16) * C + 8 * g);
    const f32x4 bp0 = *reinterpret_cast<const f32x4 *>(bpw + 4 * g), bp1 = *reinterpret_cast<const f32x4 *>(bpw + 16 + 4 * g);
    constexpr int MAXT = 8;                                     // position tiles per wave at most (TB F / 64: F <= 32)
    u32x2_conv o0[MAXT], o1[MAXT];
#pragma unroll
    for (int u = 0; u < MAXT; ++u) {
        const int i = wave + 4 * u;
        if (i < ntile) {
            const bf16x8 za = *reinterpret_cast<const bf16x8 *>(z2s + (size_t)(16 * i + r16) * ZP + 16 * g);
            const f32x4 a0 = mma16(wf0, za, bp0), a1 = mma16(wf1, za, bp1);
            o0[u] = relu_pack4(a0);
            o1[u] = relu_pack4(a1);
        }
    }
    __syncthreads();                                            // every wave has read its Z2 fragments: the tile takes the Z3 rows (dense 64-byte rows)
#pragma unroll
    for (int u = 0; u < MAXT; ++u) {
        const int i = wave + 4 * u;
        if (i < ntile) {
            *reinterpret_cast<u32x2_conv *>(z2s + (size_t)(16 * i + r16) * 64 + 8 * g) = o0[u];
            *reinterpret_cast<u32x2_conv *>(z2s + (size_t)(16 * i + r16) * 64 + 32 + 8 * g) = o1[u];
        }
    }
    __syncthreads();
    // the workgroup's rows (frames t0 .. t0 + 15 of line b, F rows each) are consecutive rows of the (B T F, C) output
    const int nvalid = min(TB, Tn - t0) * F;                    // rows of frames that exist
    bf16_t *dst = Z3 + ((size_t)b * Tn + t0) * F * C;
    for (int id = threadIdx.x; id < nvalid * 4; id += 256)
        *reinterpret_cast<bf16x8 *>(dst + (size_t)id * 8) = *reinterpret_cast<const bf16x8 *>(z2s + (size_t)id * 16);
}

// ---- frontend F1+F2, bf16 mode, C % 64 == 0: the first conv on the matrix cores ------------------------------
// Z1 = relu(conv 3x3 s2 of the 1-channel line) is a (pixels x 9) x (9 x C) product: K padded to 16 as k = 4 dt + df
// (df = 3 and dt = 3 carry zero weights), so a lane's 4 k-values are 4 CONSECUTIVE image rows of one image column --
// two 4-byte LDS reads from the transposed bf16 line tile, no im2col buffer.  v_mfma_f32_16x16x16_bf16 with the
// channels on the row side: a lane ends up with 4 consecutive channels of one Z1 pixel = one 8-byte LDS store into
// the Z1 tile [column][row][64 channels] (bf16, ReLU applied, zero outside the valid Z1 range = the second conv's
// padding).  The depthwise 3x3 s2 that follows is VALU work on that tile: a thread owns a channel pair and walks f
// with the previous Z1 row pair in registers (6 LDS reads + 9 packed FMAs per output pair).  Workgroup = 4 frames
// of one line x all channels, 64 channels per pass.  Arithmetic: bf16 pixels and conv.0 weights, fp32 accumulate;
// Z1 rounded to bf16; depthwise in fp32 (the VALU kernel above keeps everything fp32 -- fp32 mode uses that one).
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <typename TIn>
__global__ __launch_bounds__(256) void frontend_conv12_mfma_kernel(const TIn *__restrict__ X, int H, int W, int T1, int F1, int Tn, int F, int C,
                                                                   const float *__restrict__ w0, const float *__restrict__ b0,
                                                                   const float *__restrict__ w2, const float *__restrict__ b2,
                                                                   bf16_t *__restrict__ Z2, int HS, int ZR) {
    constexpr int TB = 4, NA = 2 * TB + 1, NCOL = 4 * TB + 4, ZC = 68;    // ZC: padded channel stride of the Z1 tile (conflict-free 8-byte stores)
    extern __shared__ __attribute__((aligned(16))) unsigned char c12_smem[];
    bf16_t *xs = reinterpret_cast<bf16_t *>(c12_smem);                    // [NCOL][HS]: LDS row rr <-> image row rr - 1; zero outside the image
    bf16_t *z1 = xs + NCOL * HS;                                          // [NA][ZR][ZC]: row zr <-> f1 = zr - 1
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.y, t0 = blockIdx.x * TB;
    const int col0 = 4 * t0 - 3;
    const TIn *Xb = X + (size_t)b * H * W;
    for (int i = tid; i < NCOL * HS; i += 256) {
        const int ci = i / HS, rr = i - ci * HS;
        const int w = min(max(col0 + ci, 0), W - 1), r = min(max(rr - 1, 0), H - 1);      // clamped address + select
        const float v = pixel_to_f32<TIn>(Xb[(size_t)r * W + w]);
        xs[i] = (bf16_t)((col0 + ci >= 0 && col0 + ci < W && rr >= 1 && rr <= H) ? v : 0.0f);
    }
    // rows zr = 0 (f1 = -1) and zr > F1 of the Z1 tile are the second conv's zero padding: written once
    for (int i = tid; i < NA * ZC; i += 256) {
        const int a = i / ZC, c = i - a * ZC;
        z1[(a * ZR) * ZC + c] = (bf16_t)0.0f;
        for (int zr = F1 + 1; zr < ZR; ++zr) z1[(a * ZR + zr) * ZC + c] = (bf16_t)0.0f;
    }
    __syncthreads();

    const int FT = (F1 + 15) >> 4;                         // 16-pixel tiles per Z1 column
    const int ntile = NA * FT;
    const int cp = tid & 31, grp = tid >> 5, tl = grp >> 1, fh = grp & 1;
    const int FH = (F + 1) >> 1, fbeg = fh * FH, fend = min(F, fbeg + FH);
    const int t = t0 + tl;
    for (int cb = 0; cb < C; cb += 64) {
        // ---- conv.0 on the matrix cores: wave takes pixel tiles wave, wave + 4, ...; 4 channel tiles each
        s16x4 wf[4];
        f32x4 bias[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int ch = cb + 16 * nt + r16;
            bf16x4 v = {(bf16_t)0.0f, (bf16_t)0.0f, (bf16_t)0.0f, (bf16_t)0.0f};
            if (g < 3) { v[0] = (bf16_t)w0[ch * 9 + 3 * g]; v[1] = (bf16_t)w0[ch * 9 + 3 * g + 1]; v[2] = (bf16_t)w0[ch * 9 + 3 * g + 2]; }
            wf[nt] = __builtin_bit_cast(s16x4, v);
            bias[nt] = *reinterpret_cast<const f32x4 *>(b0 + cb + 16 * nt + 4 * g);
        }
        for (int mt = wave; mt < ntile; mt += 4) {
            const int a = mt / FT, f1 = 16 * (mt - a * FT) + r16;            // this lane's pixel: Z1 column a, row f1
            const int t1 = 2 * t0 - 1 + a;
            const bf16_t *px = xs + (2 * a + g) * HS + 2 * f1;               // image rows 2 f1 - 1 .. 2 f1 + 2 of image column col0 + 2a + dt
            const unsigned lo = *reinterpret_cast<const unsigned *>(px), hi = *reinterpret_cast<const unsigned *>(px + 2);
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const s16x4 pf = __builtin_bit_cast(s16x4, (u32x2){lo, hi});
            const bool valid = t1 >= 0 && t1 < T1 && f1 < F1;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wf[nt], pf, bias[nt], 0, 0, 0);
                bf16x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = (bf16_t)(valid ? fmaxf(acc[q], 0.0f) : 0.0f);
                if (f1 + 1 < ZR) *reinterpret_cast<bf16x4 *>(z1 + (a * ZR + f1 + 1) * ZC + 16 * nt + 4 * g) = o;
            }
        }
        __syncthreads();
        // ---- depthwise 3x3 s2 over the Z1 tile: thread = channel pair cp of this pass, frame tl, half of the f range
        if (t < Tn) {
            const int c = cb + 2 * cp;
            f32x2 k2[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) k2[i] = (f32x2){w2[c * 9 + i], w2[(c + 1) * 9 + i]};
            const f32x2 bias2 = {b2[c], b2[c + 1]};
            const bf16_t *zc = z1 + (2 * tl * ZR) * ZC + 2 * cp;              // Z1 column a = 2 tl + dt, row zr = 2 f + df
            auto ld = [&](int dt, int zr) {
                const bf16x2 v = *reinterpret_cast<const bf16x2 *>(zc + (dt * ZR + zr) * ZC);
                return (f32x2){(float)v[0], (float)v[1]};
            };
            f32x2 prev[3];
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) prev[dt] = ld(dt, 2 * fbeg);
            bf16_t *out = Z2 + (((size_t)b * Tn + t) * F) * C + c;
            for (int f = fbeg; f < fend; ++f) {
                f32x2 sacc = bias2;
#pragma unroll
                for (int dt = 0; dt < 3; ++dt) {
                    const f32x2 m = ld(dt, 2 * f + 1), n = ld(dt, 2 * f + 2);
                    sacc += k2[dt * 3 + 0] * prev[dt] + k2[dt * 3 + 1] * m + k2[dt * 3 + 2] * n;
                    prev[dt] = n;
                }
                const bf16x2 o = {(bf16_t)sacc[0], (bf16_t)sacc[1]};
                *reinterpret_cast<bf16x2 *>(out + (size_t)f * C) = o;
            }
        }
        __syncthreads();
    }
}

template <typename TIn>
static inline hipError_t launch_conv12_mfma(hipStream_t s, const TIn *X, int N, int H, int W, int T1, int F1, int Tn, int F, int C, const float *w0,
                                            const float *b0, const float *w2, const float *b2, bf16_t *Z2) {
    const int FT = (F1 + 15) / 16;
    const int HS = 2 * 16 * FT + 4;                         // image rows -1 .. 32 FT + 2 (zero beyond the image)
    const int ZR = std::max(F1 + 1, 2 * F + 1) + 1;         // rows f1 = -1 .. max(F1, 2F) (zero outside [0, F1))
    const size_t lds = ((size_t)20 * HS + (size_t)9 * ZR * 68) * 2;
    hipLaunchKernelGGL((frontend_conv12_mfma_kernel<TIn>), dim3((Tn + 3) / 4, N), dim3(256), lds, s, X, H, W, T1, F1, Tn, F, C, w0, b0, w2, b2, Z2, HS, ZR);
    return hipGetLastError();
}

// ---- extra depthwise 3x3 stride-2 stage on channel-last (B,T,F,C) -> (B,T2,F2,C), bias, no activation
template <typename T>
__global__ __launch_bounds__(256) void dw3x3s2_kernel(const T *__restrict__ in, int B, int Ti, int Fi, int To, int Fo, int C,
                                                      const float *__restrict__ w, const float *__restrict__ bias, T *__restrict__ out) {
    const size_t total = (size_t)B * To * Fo * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int f = (int)(r % Fo); r /= Fo;
        const int t = (int)(r % To);
        const int b = (int)(r / To);
        float s = bias[c];
#pragma unroll
        for (int dt = 0; dt < 3; ++dt)
#pragma unroll
            for (int df = 0; df < 3; ++df) {
                const int ti = 2 * t + dt - 1, fi = 2 * f + df - 1;
                if (ti >= 0 && ti < Ti && fi >= 0 && fi < Fi)
                    s += w[c * 9 + dt * 3 + df] * to_f32(in[(((size_t)b * Ti + ti) * Fi + fi) * C + c]);
            }
        out[i] = from_f32<T>(s);
    }
}

// ---- conv module: depthwise k-tap FIR along time + folded BatchNorm + SiLU ----------------------
//   u[b][t][c] = silu(bias'[c] + sum_tau w'[tau][c] g[b][t + tau - (k-1)/2][c]),  zero outside [0,T) of the
//   PADDED batch (the reference pads the batch tensor, not the individual line: SURVEY A.1b).
// Workgroup = TT output frames x 256 channels of one line; the TT+k-1 input frames are staged in LDS
// (coalesced over channels); each of 128 threads owns two adjacent channels.  HBM-bound: algorithmic
// bytes = one read + one write of the (B,T,D) operand.

template <typename T, int TT>
__global__ __launch_bounds__(128) void dwconv_bn_silu_kernel(const T *__restrict__ g, int Tn, int D, int k,
                                                             const float *__restrict__ w, const float *__restrict__ bias,
                                                             T *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dw_smem[];
    typedef typename Pair<T>::type P2;
    P2 *win = reinterpret_cast<P2 *>(dw_smem);                 // [TT + k - 1][128] channel pairs
    const int b = blockIdx.y, t0 = blockIdx.x * TT, c0 = blockIdx.z * 256;
    const int pad = (k - 1) / 2, rows = TT + k - 1;
    const int c = c0 + 2 * threadIdx.x;
    const bool live = c < D;                                   // D is even
    const T *gb = g + (size_t)b * Tn * D;
    for (int r = 0; r < rows; ++r) {
        const int t = t0 + r - pad;
        P2 v; v[0] = (T)0.0f; v[1] = (T)0.0f;
        if (live && t >= 0 && t < Tn) v = *reinterpret_cast<const P2 *>(gb + (size_t)t * D + c);
        win[r * 128 + threadIdx.x] = v;
    }
    // each thread reads back only what it wrote itself: no barrier needed
    if (!live) return;
    float a0[TT], a1[TT];
    const float bb0 = bias[c], bb1 = bias[c + 1];
#pragma unroll
    for (int i = 0; i < TT; ++i) { a0[i] = bb0; a1[i] = bb1; }
    for (int tau = 0; tau < k; ++tau) {
        const float w0 = w[(size_t)tau * D + c], w1 = w[(size_t)tau * D + c + 1];
#pragma unroll
        for (int i = 0; i < TT; ++i) {
            const P2 v = win[(i + tau) * 128 + threadIdx.x];
            a0[i] += w0 * to_f32(v[0]);
            a1[i] += w1 * to_f32(v[1]);
        }
    }
#pragma unroll
    for (int i = 0; i < TT; ++i) {
        const int t = t0 + i;
        if (t < Tn) {
            P2 o; o[0] = from_f32<T>(silu_f(a0[i])); o[1] = from_f32<T>(silu_f(a1[i]));
            *reinterpret_cast<P2 *>(out + ((size_t)b * Tn + t) * D + c) = o;
        }
    }
}

// Compile-time-k variant: the TT + K - 1 input frames of the thread's channel pair are loaded up front (all
// loads in flight together), taps and accumulators live in registers, the (frame, tap) loops are fully
// unrolled -- no LDS.  Same arithmetic order per output as the generic kernel (taps ascending).
template <typename T, int K, int TT>
__global__ __launch_bounds__(128) void dwconv_bn_silu_k_kernel(const T *__restrict__ g, int Tn, int D,
                                                               const float *__restrict__ w, const float *__restrict__ bias,
                                                               T *__restrict__ out) {
    typedef typename Pair<T>::type P2;
    constexpr int PAD = (K - 1) / 2, ROWS = TT + K - 1;
    const int b = blockIdx.y, t0 = blockIdx.x * TT, c = blockIdx.z * 256 + 2 * threadIdx.x;
    if (c >= D) return;
    const T *gb = g + (size_t)b * Tn * D + c;
    P2 xin[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int t = t0 + r - PAD;
        // clamped address + select (a branch around each load would serialise the ROWS loads)
        P2 v = *reinterpret_cast<const P2 *>(gb + (size_t)min(max(t, 0), Tn - 1) * D);
        if (t < 0 || t >= Tn) { v[0] = (T)0.0f; v[1] = (T)0.0f; }
        xin[r] = v;
    }
    float w0[K], w1[K];
#pragma unroll
    for (int tau = 0; tau < K; ++tau) {
        const f32x2 ww = *reinterpret_cast<const f32x2 *>(w + (size_t)tau * D + c);
        w0[tau] = ww[0]; w1[tau] = ww[1];
    }
    const f32x2 bb = *reinterpret_cast<const f32x2 *>(bias + c);
#pragma unroll
    for (int i = 0; i < TT; ++i) {
        float a0 = bb[0], a1 = bb[1];
#pragma unroll
        for (int tau = 0; tau < K; ++tau) {
            a0 += w0[tau] * to_f32(xin[i + tau][0]);
            a1 += w1[tau] * to_f32(xin[i + tau][1]);
        }
        const int t = t0 + i;
        if (t < Tn) {
            P2 o; o[0] = from_f32<T>(silu_f(a0)); o[1] = from_f32<T>(silu_f(a1));
            *reinterpret_cast<P2 *>(out + ((size_t)b * Tn + t) * D + c) = o;
        }
    }
}

template <typename T>
static inline void launch_dwconv(hipStream_t s, const T *g, int N, int Tn, int D, int k, const float *w, const float *bias, T *out) {
    constexpr int TT = 16;
    dim3 grid(ceil_div(Tn, TT), N, ceil_div(D, 256));
#define DWK(KK) hipLaunchKernelGGL((dwconv_bn_silu_k_kernel<T, KK, TT>), grid, dim3(128), 0, s, g, Tn, D, w, bias, out)
    switch (k) {
        case 7: DWK(7); break;
        case 9: DWK(9); break;
        case 15: DWK(15); break;
        case 31: DWK(31); break;
        default: {
            const size_t lds = (size_t)(TT + k - 1) * 128 * 2 * sizeof(T);
            hipLaunchKernelGGL((dwconv_bn_silu_kernel<T, TT>), grid, dim3(128), lds, s, g, Tn, D, k, w, bias, out);
        }
    }
#undef DWK
}
