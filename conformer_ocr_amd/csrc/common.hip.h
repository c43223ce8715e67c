// Shared device helpers for the gfx950 kernels: element types, MFMA wrappers, reductions.
//
// Every matrix product in this library is written against ONE fragment shape so that the bf16 and
// the fp32 compute modes share their indexing: a "k-chunk" is 32 consecutive k; lane l of a wave
// (r = l & 15, g = l >> 4) holds the 8 elements k = 8g .. 8g+7 of row r of A (or of column r of B).
//   bf16: one v_mfma_f32_16x16x32_bf16 consumes a k-chunk (operand map of the ISA: A[r][8g+j]).
//   fp32: eight v_mfma_f32_16x16x4_f32; MFMA j takes element j of every lane, i.e. k = 8g+j for
//         g = 0..3 -- a permutation of k inside the chunk, the same for A and B, so the sum is the
//         same.  Exact fp32 products, fp32 accumulate.
// The 16x16 accumulator map is dtype independent: column = l & 15, rows 4g .. 4g+3 in regs 0..3.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((address_space(3))) void *lds_ptr_t;            // operands of __builtin_amdgcn_global_load_lds (LDS-DMA)
typedef __attribute__((address_space(1))) const void *gbl_ptr_t;

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

#define COCR_WAVE 64

template <typename T> struct FragOf;
template <> struct FragOf<bf16_t> { typedef bf16x8 type; };
template <> struct FragOf<float> { typedef f32x8 type; };

// one k-chunk (32 k) of a 16x16 output tile
__device__ __forceinline__ f32x4 mma16(const bf16x8 &a, const bf16x8 &b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma16(const f32x8 &a, const f32x8 &b, f32x4 c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
    return c;
}

// 8 consecutive elements (16 B bf16 / 32 B fp32), pointer aligned to its size
__device__ __forceinline__ bf16x8 load_frag(const bf16_t *p) { return *reinterpret_cast<const bf16x8 *>(p); }
__device__ __forceinline__ f32x8 load_frag(const float *p) {   // two 16-byte accesses: p is only 16-byte aligned in LDS
    const f32x4 lo = *reinterpret_cast<const f32x4 *>(p), hi = *reinterpret_cast<const f32x4 *>(p + 4);
    return (f32x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
template <typename T> __device__ __forceinline__ typename FragOf<T>::type zero_frag() {
    typename FragOf<T>::type z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (T)0.0f;
    return z;
}

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x) { return (T)x; }   // bf16: round to nearest even

// v_rcp_f32 (1 ulp) instead of an IEEE division (10 instructions): SiLU / sigmoid sit in GEMM epilogues
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }

// Wave-wide reductions by DPP (data-parallel primitives: a lane permutation folded into the VALU op), result
// uniform in all 64 lanes.  __shfl_xor compiles to ds_bpermute (an LDS-crossbar round trip per step: a 6-step
// dependent chain cost ~500 cycles per reduction in the LayerNorm epilogues).  Steps: xor 1, xor 2 (quad_perm),
// half-row mirror, row mirror -> every lane holds its 16-lane row's result; row_bcast15 into rows 1,3, row_bcast31
// into rows 2,3 -> lane 63 holds the wave's result; v_readlane broadcasts it.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_f32(float v, float old) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f32<0xB1, 0xF>(v, 0.f);      // quad_perm(1,0,3,2)
    v += dpp_f32<0x4E, 0xF>(v, 0.f);      // quad_perm(2,3,0,1)
    v += dpp_f32<0x141, 0xF>(v, 0.f);     // row_half_mirror
    v += dpp_f32<0x140, 0xF>(v, 0.f);     // row_mirror
    v += dpp_f32<0x142, 0xA>(v, 0.f);     // row_bcast15 -> rows 1, 3
    v += dpp_f32<0x143, 0xC>(v, 0.f);     // row_bcast31 -> rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f32<0xB1, 0xF>(v, v));
    v = fmaxf(v, dpp_f32<0x4E, 0xF>(v, v));
    v = fmaxf(v, dpp_f32<0x141, 0xF>(v, v));
    v = fmaxf(v, dpp_f32<0x140, 0xF>(v, v));
    v = fmaxf(v, dpp_f32<0x142, 0xA>(v, v));
    v = fmaxf(v, dpp_f32<0x143, 0xC>(v, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }
