// Line pre-processing in front of the path (kraken ImageInputTransforms(1, 96, 0, 1, (16, 0)) -- grayscale, scale to
// height 96 keeping the aspect ratio, pad 16 px left and right): raw 8-bit line crops (grayscale or RGB, any height) -> the
// (N, 96, W) u8 batch the forward ingests (pixel / 255 = the reference's [0, 1] floats).
//
// The resize is Pillow's 8-bit LANCZOS resampler, integer for integer (oracle/preproc_ref.py, pinned bit-exactly against
// Pillow): per output pixel a window of taps in 22-bit fixed point, horizontal pass first into a u8 intermediate, then the
// vertical pass, each value clip8((2^21 + sum pixel * tap) >> 22).  The tap tables depend only on (in size, out size): they
// are built on the host in double precision exactly like Pillow's precompute_coeffs / normalize_coeffs_8bpc (same libm sin)
// and shipped with the call.  Both passes are byte work bound by HBM: one thread per output pixel, taps read from
// neighbouring bytes (L1 / L2 hits), lanes along the row so that loads and stores coalesce.
#pragma once
#include <cmath>
#include <vector>

#include "common.hip.h"

struct PreLine {
    long long in_off;      // first byte of the line in the packed pixel buffer
    long long tmp_off;     // first byte of its (h, ow) intermediate
    int h, w, cpp, ow;     // input rows, columns, bytes per pixel (1 | 3), scaled width
    int hb, hk, hks;       // horizontal pass: offsets (in ints) of bounds[ow][2] / taps[ow][hks] in the table buffer, taps per pixel
    int vb, vk, vks;       // vertical pass: bounds[out_h][2] / taps[out_h][vks]
};

__device__ __forceinline__ int pre_pixel(const unsigned char *row, int x, int cpp) {
    if (cpp == 1) return row[x];
    const unsigned char *p = row + 3 * x;                  // Pillow's RGB -> L
    return (p[0] * 19595 + p[1] * 38470 + p[2] * 7471 + 0x8000) >> 16;
}
__device__ __forceinline__ int pre_clip8(int acc) { return min(max(acc >> 22, 0), 255); }

__global__ __launch_bounds__(256) void preproc_h_kernel(const unsigned char *__restrict__ pixels, const PreLine *__restrict__ lines,
                                                        const int *__restrict__ tab, unsigned char *__restrict__ tmp) {
    const PreLine L = lines[blockIdx.z];
    const int y = blockIdx.y, xx = blockIdx.x * 256 + threadIdx.x;
    if (y >= L.h || xx >= L.ow) return;
    const unsigned char *row = pixels + L.in_off + (size_t)y * L.w * L.cpp;
    const int x0 = tab[L.hb + 2 * xx], n = tab[L.hb + 2 * xx + 1];
    const int *k = tab + L.hk + (size_t)xx * L.hks;
    int acc = 1 << 21;
    for (int t = 0; t < n; ++t) acc += pre_pixel(row, x0 + t, L.cpp) * k[t];
    tmp[L.tmp_off + (size_t)y * L.ow + xx] = (unsigned char)pre_clip8(acc);
}

__global__ __launch_bounds__(256) void preproc_v_kernel(const unsigned char *__restrict__ tmp, const PreLine *__restrict__ lines,
                                                        const int *__restrict__ tab, unsigned char *__restrict__ out, int out_h, int out_w, int pad) {
    const PreLine L = lines[blockIdx.z];
    const int yy = blockIdx.y, xc = blockIdx.x * 256 + threadIdx.x;
    if (xc >= out_w) return;
    int v = 0;                                             // left / right padding and the batch's right fill
    const int xx = xc - pad;
    if (xx >= 0 && xx < L.ow) {
        const int y0 = tab[L.vb + 2 * yy], n = tab[L.vb + 2 * yy + 1];
        const int *k = tab + L.vk + (size_t)yy * L.vks;
        const unsigned char *col = tmp + L.tmp_off + (size_t)y0 * L.ow + xx;
        int acc = 1 << 21;
        for (int t = 0; t < n; ++t) acc += (int)col[(size_t)t * L.ow] * k[t];
        v = pre_clip8(acc);
    }
    out[((size_t)blockIdx.z * out_h + yy) * out_w + xc] = (unsigned char)v;
}

// ---- host: Pillow's tap tables ---------------------------------------------------------------------------------
static inline double pre_lanczos(double x) {
    if (-3.0 <= x && x < 3.0) {
        if (x == 0.0) return 1.0;
        const double a = x * M_PI, b = x / 3.0 * M_PI;
        return (sin(a) / a) * (sin(b) / b);
    }
    return 0.0;
}
// appends bounds[out][2] and taps[out][ksize] to `tab`; returns their offsets and ksize
static inline void pre_coeffs(int in_size, int out_size, std::vector<int> &tab, int *b_off, int *k_off, int *ksize_out) {
    const double scale = (double)in_size / out_size, fscale = scale < 1.0 ? 1.0 : scale, support = 3.0 * fscale, ss = 1.0 / fscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    *b_off = (int)tab.size();
    tab.resize(tab.size() + (size_t)out_size * 2);
    *k_off = (int)tab.size();
    tab.resize(tab.size() + (size_t)out_size * ksize, 0);
    *ksize_out = ksize;
    std::vector<double> w((size_t)ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) { w[x] = pre_lanczos((x + xmin - center + 0.5) * ss); ww += w[x]; }
        int *k = tab.data() + *k_off + (size_t)xx * ksize;
        for (int x = 0; x < xmax; ++x) {
            const double v = ww != 0.0 ? w[x] / ww : w[x];
            k[x] = v < 0 ? (int)(-0.5 + v * (double)(1 << 22)) : (int)(0.5 + v * (double)(1 << 22));
        }
        tab[*b_off + 2 * xx] = xmin;
        tab[*b_off + 2 * xx + 1] = xmax;
    }
}
static inline int pre_scaled_width(int h, int w, int out_h) {
    const int ow = (int)((double)w * out_h / h);
    return ow < 1 ? 1 : ow;
}
