// Shared device helpers for the gfx950 kernels.  Wavefront = 64 lanes throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ligdiff.h"

#define LD_H 256          // hidden width
#define LD_A 119          // atom feature width
#define LD_R 32           // radial basis width
#define LD_P 15           // property width
#define LD_HEADS 8
#define LD_DH 32

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define LD_CHECK_LAUNCH()                                   \
  do {                                                      \
    if (hipGetLastError() != hipSuccess) return LIGDIFF_ERR_LAUNCH; \
  } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));

// erf with absolute error <= 3e-7 (Abramowitz & Stegun 7.1.28):
//   1 - (1 + a1 t + ... + a6 t^6)^-16,  t = |x|.
// Six FMAs, four squarings and ONE reciprocal, no exponential: the transcendental unit runs at
// a quarter of the FMA rate, and GELU sits inside GEMM epilogues that are VALU-bound.  The
// two-wide form keeps the polynomial on packed fp32 instructions (v_pk_fma_f32: two lanes of
// work per issue).  For |x| > ~9 the power overflows to +inf and the reciprocal gives 0.
__device__ __forceinline__ f32x2 erf_fast2(f32x2 x) {
  const f32x2 t = __builtin_elementwise_abs(x);
  f32x2 p = t * 0.0000430638f + 0.0002765672f;
  p = p * t + 0.0001520143f;
  p = p * t + 0.0092705272f;
  p = p * t + 0.0422820123f;
  p = p * t + 0.0705230784f;
  p = p * t + 1.0f;
  p = p * p; p = p * p; p = p * p; p = p * p;
  f32x2 r;
  r[0] = 1.0f - __builtin_amdgcn_rcpf(p[0]);
  r[1] = 1.0f - __builtin_amdgcn_rcpf(p[1]);
  r[0] = copysignf(r[0], x[0]);
  r[1] = copysignf(r[1], x[1]);
  return r;
}
// nn.GELU() default (erf form), model1.py:602 / :615 / :638
__device__ __forceinline__ f32x2 gelu_exact2(f32x2 v) {
  const f32x2 hv = v * 0.5f;
  return hv * erf_fast2(v * 0.70710678118654752440f) + hv;
}
__device__ __forceinline__ float gelu_exact(float v) {
  const f32x2 g = gelu_exact2(f32x2{v, v});
  return g[0];
}
// SiLU through the hardware exp2 / reciprocal (relative error ~1e-6 for |v| < 16): it sits
// in GEMM epilogues, where the IEEE division and the library expf cost ~4x the instructions.
__device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ float softplus(float v) { return v > 20.0f ? v : log1pf(expf(v)); }

// d/dv of the activations above (used by the backward kernels)
__device__ __forceinline__ float gelu_exact_grad(float v) {
  const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * v * v);
  return cdf + v * pdf;
}
__device__ __forceinline__ float silu_grad(float v) {
  const float s = 1.0f / (1.0f + expf(-v));
  return s * (1.0f + v * (1.0f - s));
}

// Lane exchanges on the VALU data-parallel-primitive path (no LDS crossbar traffic).
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
// v + (v of lane ^ 16): v_permlane16_swap exchanges the odd 16-lane rows of its first
// operand with the even rows of its second, so swapping a value with its own copy leaves
// {row0,row0,row2,row2} and {row1,row1,row3,row3}.
__device__ __forceinline__ float add_xor16(float v) {
  const auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(p[0]) + __uint_as_float(p[1]);
}
// Sum over the 32 lanes of each half-wave (lanes 0-31 and 32-63 reduce separately); every
// lane of the half receives the total.
__device__ __forceinline__ float half_wave_sum(float v) {
  v += dpp_move<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);   // row_half_mirror: lane i <-> 7-i of each 8
  v += dpp_move<0x140>(v);   // row_mirror: lane i <-> 15-i of each 16
  return add_xor16(v);
}
__device__ __forceinline__ float wave_sum(float v) {
  v += __shfl_xor(v, 32);
  return half_wave_sum(v);
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 32));
  v = fmaxf(v, __shfl_xor(v, 16));
  v = fmaxf(v, __shfl_xor(v, 8));
  v = fmaxf(v, __shfl_xor(v, 4));
  v = fmaxf(v, __shfl_xor(v, 2));
  v = fmaxf(v, __shfl_xor(v, 1));
  return v;
}

// ------------------------------------------------------------------ Philox4x32-10 and dropout masks
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

// One dropout site of a training forward / backward pass (model1.py:263-274: attention probabilities,
// FFN hidden after GELU, FFN output).  An element is KEPT iff its 32 random bits are >= thr
// (thr = p 2^32, so thr == 0 means no dropout) and then scaled by inv_keep = 1 / (1 - p), as
// torch.nn.functional.dropout does.  The bits of an element are a pure function of (seed, site,
// global element index): forward and backward regenerate the same mask, and it does not depend on
// how the batch is cut into shards (row_offset = global index of the call's first atom).
struct DropSite {
  uint64_t seed;
  int64_t row_offset;
  uint32_t site;
  uint32_t thr;
  float inv_keep;
};
#define LD_DROP_STREAM 0xD20F0000u
// Scales of the four elements of quad `quad` of a site.
__device__ __forceinline__ f32x4 drop_scale4(const DropSite& d, uint64_t quad) {
  uint32_t c[4] = {(uint32_t)quad, (uint32_t)(quad >> 32), d.site, LD_DROP_STREAM};
  philox4x32_10(c, (uint32_t)d.seed, (uint32_t)(d.seed >> 32));
  return f32x4{c[0] >= d.thr ? d.inv_keep : 0.f, c[1] >= d.thr ? d.inv_keep : 0.f,
               c[2] >= d.thr ? d.inv_keep : 0.f, c[3] >= d.thr ? d.inv_keep : 0.f};
}
// Row sites ([rows, width] activations, width a multiple of 128): the quad of element (row, col) holds
// the four columns {b, b + 32, b + 64, b + 96} of its 128-column block - what one lane of a GEMM
// epilogue owns - and the element is component (col >> 5) & 3 of it.
__device__ __forceinline__ uint64_t drop_row_quad(const DropSite& d, int row, int col, int width) {
  return (uint64_t)(d.row_offset + row) * (uint64_t)(width >> 2) + (uint64_t)((col >> 7) * 32 + (col & 31));
}
__device__ __forceinline__ float drop_row_scale(const DropSite& d, int row, int col, int width) {
  return drop_scale4(d, drop_row_quad(d, row, col, width))[(col >> 5) & 3];
}
// Attention site: element (query atom, head, key index j within the molecule) -> quad of four
// consecutive keys; the element is component j & 3.
__device__ __forceinline__ uint64_t drop_att_quad(const DropSite& d, int q_atom, int head, int j) {
  return ((uint64_t)(d.row_offset + q_atom) * 8u + (uint64_t)head) * 32u + (uint64_t)(j >> 2);
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
