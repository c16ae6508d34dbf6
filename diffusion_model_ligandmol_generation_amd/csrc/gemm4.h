// Column-split row-slab GEMM with fp32 operands multiplied on the bf16 matrix pipe.
//
// An fp32 number is the exact sum of three bf16 numbers (8 + 8 + 8 significand bits, obtained by
// truncating and subtracting twice), so  a b = sum_{i,j} a_i b_j  with nine exact bf16 x bf16
// products.  The kernel issues the six largest of them (i + j <= 2) as v_mfma_f32_32x32x16_bf16
// with fp32 accumulation; the three dropped products are below 2^-23 |a b|, the size of one fp32
// rounding.  Six of these instructions cost 192 cycles per 32 x 32 x 16 block against 512 for eight
// v_mfma_f32_32x32x2_f32, so the main loop has 2.67x the fp32 pipe's ceiling.
//
// Work split, accumulator layout and epilogues are those of gemm_cs_kernel (gemm3.h): eight
// waves, wave = 32-row slab x column half, acc[NTW] of 32 x 32, EPI::cs.  What differs:
//   * W arrives PRE-SPLIT (k_split_weight below): per column block and 16-deep K chunk one
//     contiguous piece  [plane 3][lane half 2][n 64 NTW][8 bf16]  that LDS-DMA copies linearly and
//     whose fragment reads (one ds_read_b128 per plane and column tile) are dense 512-B runs,
//     hence free of bank conflicts.  The 8 values of lane half h are k = 4h..4h+3, 8+4h..8+4h+3:
//     the same k set the A fragment reads below deliver.
//   * A is still fp32 in LDS (same DMA, same swizzle, same two ds_read_b128 per chunk as gemm3.h)
//     and is split in registers: 11 VALU instructions per pair of values.
// K must be a multiple of 16; rows beyond M are clamped on load; W rows beyond n_valid are zero
// in the split copy.
#pragma once
#include <stdint.h>

#include "gemm3.h"

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// bf16 elements of one K chunk of a column block of 64 NTW columns
#define G4_CHUNK_ELEMS(NTW) (3 * 2 * 64 * (NTW) * 8)   // = 48 bf16 per column

struct SplitW {
  const uint16_t* blk[8];   // blk[by]: [K/16 chunks][3 planes][2 lane halves][64 NTW columns][8] bf16
};

// Truncating three-way split of two floats into packed bf16 pairs (low half = x).
__device__ __forceinline__ void split_pair(float x, float y, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
  const uint32_t xb = __float_as_uint(x), yb = __float_as_uint(y);
  const float rx = x - __uint_as_float(xb & 0xffff0000u);
  const float ry = y - __uint_as_float(yb & 0xffff0000u);
  const uint32_t rxb = __float_as_uint(rx), ryb = __float_as_uint(ry);
  const float sx = rx - __uint_as_float(rxb & 0xffff0000u);
  const float sy = ry - __uint_as_float(ryb & 0xffff0000u);
  p0 = __builtin_amdgcn_perm(yb, xb, 0x07060302u);
  p1 = __builtin_amdgcn_perm(ryb, rxb, 0x07060302u);
  p2 = __builtin_amdgcn_perm(__float_as_uint(sy), __float_as_uint(sx), 0x07060302u);
}

// W[n_valid rows, K] (row stride ldw) -> split copy of one column block (layout above).
// grid = (ceil(ncols * K/16 / 256)), ncols = 64 NTW; one thread per (chunk, column).
static __global__ void k_split_weight(const float* __restrict__ W, int ldw, int n_valid, int K, int ncols,
                               uint16_t* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int nk = K / 16;
  if (idx >= nk * ncols) return;
  const int kc = idx / ncols, n = idx - kc * ncols;
  float v[16];
  if (n < n_valid) {
    const f32x4* src = reinterpret_cast<const f32x4*>(W + (size_t)n * ldw + kc * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 t = src[q];
      v[q * 4 + 0] = t[0]; v[q * 4 + 1] = t[1]; v[q * 4 + 2] = t[2]; v[q * 4 + 3] = t[3];
    }
  } else {
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = 0.f;
  }
  uint16_t* chunk = out + (size_t)kc * (3 * 2 * ncols * 8);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    u32x4 p[3];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // element e = 2j, 2j+1 of lane half h:  k = 4h + e (e < 4),  8 + 4h + (e - 4) otherwise
      const int k0 = (j < 2) ? 4 * h + 2 * j : 8 + 4 * h + 2 * (j - 2);
      uint32_t a, b, c;
      split_pair(v[k0], v[k0 + 1], a, b, c);
      p[0][j] = a; p[1][j] = b; p[2][j] = c;
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      *reinterpret_cast<u32x4*>(chunk + ((size_t)(pl * 2 + h) * ncols + n) * 8) = p[pl];
  }
}

// Transposing variant for input-gradient GEMMs  dX = dY W  (W [K, ncols], row stride ldw): the B
// operand is W^T, so element k of output column n is W[k, n].  One launch converts every 256-column
// block; block b lands at out + b * (K / 16) * G4_CHUNK_ELEMS(4).  One thread per (block, chunk,
// column), columns fastest, so the 16 strided loads of a thread are coalesced across the wave.
static __global__ void k_split_weight_t(const float* __restrict__ W, int ldw, int K, int ncols,
                                        uint16_t* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int nk = K / 16;
  if (idx >= nk * ncols) return;
  const int n_all = idx % ncols, kc = idx / ncols;
  const int b = n_all >> 8, n = n_all & 255;
  float v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = W[(size_t)(kc * 16 + j) * ldw + n_all];
  uint16_t* chunk = out + ((size_t)b * nk + kc) * G4_CHUNK_ELEMS(4);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    u32x4 p[3];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k0 = (j < 2) ? 4 * h + 2 * j : 8 + 4 * h + 2 * (j - 2);
      uint32_t a, bb, c;
      split_pair(v[k0], v[k0 + 1], a, bb, c);
      p[0][j] = a; p[1][j] = bb; p[2][j] = c;
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      *reinterpret_cast<u32x4*>(chunk + ((size_t)(pl * 2 + h) * 256 + n) * 8) = p[pl];
  }
}

// Stage buffers: W chunks (6 NTW KiB, L2-resident) are double-buffered; A chunks (8 KiB) have NSA
// buffers.  With NSA = 3 a wave issues its W pieces of chunk kc + 1 and THEN its A piece of chunk
// kc + 2, so "all but the newest DMA done" (s_waitcnt vmcnt(1)) is exactly "chunk kc + 1 landed";
// measured on the FFN shapes this is 2-3 % slower than plain double buffering (NSA = 2, the default):
// with four waves per SIMD the DMA latency is already covered.
//
// WAVES = 8: the column-split form above (EPI::cs).  WAVES = 4: wave = 32-row slab x ALL 32 NTW columns of
// the block, the ownership of gemm_rows2_kernel, for epilogues that need whole rows in one wave
// (EPI::operator()); 256 threads, no mailbox.
template <class AP, class EPI, int NTW = 4, int NSA = 2, int WAVES = 8>
__global__ __launch_bounds__(64 * WAVES, 2) void gemm_cs_split_kernel(GemmCore g, SplitW sw, AP ap, EPI epi) {
  constexpr int NCOL = (WAVES == 8) ? 64 * NTW : 32 * NTW;   // columns of the block
  constexpr int A_FL = 128 * G2_ROWF;                 // fp32 A stage: 128 rows x 16
  constexpr int CHUNK = 3 * 2 * NCOL * 8;             // bf16 elements of one split W chunk
  constexpr int B_FL = CHUNK / 2;                     // split W stage, in floats
  constexpr int PIECES = B_FL / 256;                  // 1-KiB DMA pieces of the W stage
  constexpr int PW = (PIECES + WAVES - 1) / WAVES;    // per wave (the last waves repeat the last piece)
  constexpr int PA = 8 / WAVES;                       // A pieces (16 rows each) per wave
  constexpr int PLANE_B = 2 * NCOL * 16;              // bytes per plane
  __shared__ __attribute__((aligned(1024))) float smem4[NSA * A_FL + 2 * B_FL + (WAVES == 8 ? 4 * 256 : 0)];
  float* const sA_base = smem4;
  float* const sB_base = smem4 + NSA * A_FL;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slab = wave & 3, half = (WAVES == 8) ? wave >> 2 : 0;
  const int by = blockIdx.y;
  const int row_blk = blockIdx.x * 128;

  f32x16 acc[NTW];
#pragma unroll
  for (int ct = 0; ct < NTW; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;

  const int pr = lane >> 2, ps = lane & 3;
  const int lslot = ps ^ ((pr >> 2) & 3);
  int a_row[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int r = row_blk + (wave * PA + i) * 16 + pr;
    a_row[i] = r < g.M ? r : g.M - 1;
  }
  const uint16_t* bsrc[PW];
  int bdst[PW];
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    int p = wave * PW + i;
    p = p < PIECES ? p : PIECES - 1;
    bsrc[i] = sw.blk[by] + p * 512 + lane * 8;
    bdst[i] = p * 256;
  }
  auto issue_a = [&](int kc, int buf) {
#pragma unroll
    for (int i = 0; i < PA; ++i)
      lds_dma16(ap.ptr(a_row[i], kc * G2_BK + lslot * 4), sA_base + buf * A_FL + (wave * PA + i) * 256);
  };
  auto issue_b = [&](int kc, int buf) {
#pragma unroll
    for (int i = 0; i < PW; ++i)
      lds_dma16(reinterpret_cast<const float*>(bsrc[i] + (size_t)kc * CHUNK),
                sB_base + buf * B_FL + bdst[i]);
  };

  const int nk = g.K / G2_BK;
  const int fi = lane & 31, hk = lane >> 5;
  const int swz = (fi >> 2) & 3;
  const int off0 = ((0 * 2 + hk) ^ swz) * 4, off1 = ((1 * 2 + hk) ^ swz) * 4;
  const int b_lane = (hk * NCOL + half * 32 * NTW + fi) * 16;     // byte offset inside a plane
  issue_a(0, 0);
  issue_b(0, 0);
  if (NSA == 3 && nk > 1) issue_a(1, 1);
  int cur_a = 0;
  for (int kc = 0; kc < nk; ++kc) {
    if (NSA == 3 && kc + 1 < nk) wait_vm_count<PA>();
    else wait_vm_count<0>();
    __builtin_amdgcn_s_barrier();
    if (kc + 1 < nk) issue_b(kc + 1, (kc + 1) & 1);
    if (kc + NSA - 1 < nk) issue_a(kc + NSA - 1, cur_a == 0 ? NSA - 1 : cur_a - 1);
    const float* sA = sA_base + cur_a * A_FL;
    const char* sB = reinterpret_cast<const char*>(sB_base + (kc & 1) * B_FL) + b_lane;
    const float* arow = sA + (slab * 32 + fi) * G2_ROWF;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(arow + off0);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(arow + off1);
    u32x4 b[2][3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) b[0][pl] = *reinterpret_cast<const u32x4*>(sB + pl * PLANE_B);
    u32x4 as[3];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      uint32_t p0, p1, p2;
      split_pair(a0[2 * j], a0[2 * j + 1], p0, p1, p2);
      as[0][j] = p0; as[1][j] = p1; as[2][j] = p2;
      split_pair(a1[2 * j], a1[2 * j + 1], p0, p1, p2);
      as[0][2 + j] = p0; as[1][2 + j] = p1; as[2][2 + j] = p2;
    }
    const bf16x8 A0 = __builtin_bit_cast(bf16x8, as[0]);
    const bf16x8 A1 = __builtin_bit_cast(bf16x8, as[1]);
    const bf16x8 A2 = __builtin_bit_cast(bf16x8, as[2]);
#pragma unroll
    for (int ct = 0; ct < NTW; ++ct) {
      const bf16x8 B0 = __builtin_bit_cast(bf16x8, b[ct & 1][0]);
      const bf16x8 B1 = __builtin_bit_cast(bf16x8, b[ct & 1][1]);
      const bf16x8 B2 = __builtin_bit_cast(bf16x8, b[ct & 1][2]);
      // smallest products first
      acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, B0, acc[ct], 0, 0, 0);
      acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B2, acc[ct], 0, 0, 0);
      acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1, acc[ct], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (ct + 1 < NTW) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          b[(ct + 1) & 1][pl] = *reinterpret_cast<const u32x4*>(sB + pl * PLANE_B + (ct + 1) * 32 * 16);
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B0, acc[ct], 0, 0, 0);
      acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B1, acc[ct], 0, 0, 0);
      acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, acc[ct], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    cur_a = cur_a == NSA - 1 ? 0 : cur_a + 1;
  }
  if constexpr (WAVES == 8) {
    HalfMailbox mb{smem4 + NSA * A_FL + 2 * B_FL, wave, lane};
    epi.cs(acc, row_blk + slab * 32, by, half, lane, mb);
  } else {
    epi(acc, row_blk + wave * 32, by, lane);
  }
}
