// Column-split row-slab GEMM for 256-wide outputs: C[M, 256 * gridDim.y] = epilogue(A W^T).
//
// A workgroup still owns 128 rows x 256 columns, but runs EIGHT waves: wave w multiplies row
// slab (w & 3) by column half (w >> 2), i.e. a 32 x 128 tile held as four accumulators of
// v_mfma_f32_32x32x2_f32.  The 4-wave kernels of gemm.h / gemm2.h put ONE wave on each SIMD
// when M / 128 is about the number of CUs (M = 29 696 atoms -> 232 workgroups on 256 CUs), so
// every LDS read, barrier and epilogue load is exposed; here a SIMD holds two to four waves
// and the halves of a row meet only in the epilogue:
//   * row statistics (LayerNorm mean / variance, dot products) are computed per half and
//     combined through a 4-KiB LDS mailbox with one workgroup barrier.  Variances combine by
//     the pairwise update  M2 = M2a + M2b + (na nb / n) (mean_a - mean_b)^2  (na = nb = 128).
//   * Operands arrive by LDS-DMA exactly as in gemm2.h (16-deep K chunks, XOR-swizzled 64-B
//     rows) but through THREE stage buffers with prefetch distance two: a chunk is only 32
//     MFMAs per wave, shorter than the L2 latency, so the wait before the barrier leaves the
//     newest chunk in flight (s_waitcnt vmcnt(3): each wave issues three DMAs per chunk).
// K must be a multiple of 16; rows beyond M and W rows beyond n_valid are clamped on load.
#pragma once
#include "gemm2.h"

#define G3_STAGE_FLOATS ((128 + 256) * G2_ROWF)               // 24 KiB per stage
#define G3_LDS_BYTES ((3 * G3_STAGE_FLOATS + 4 * 256) * 4)     // stages + mailbox

// Mailbox for per-row partial results of the two column halves of a slab.  Area a holds
// [8 waves][2 lane halves][16 registers]; a value must already be uniform over the 32 lanes
// of a half-wave (half_wave_sum output).
struct HalfMailbox {
  float* base; int wave; int lane;
  __device__ __forceinline__ void put(int area, const float (&v)[16]) const {
    float* p = base + area * 256 + wave * 32 + (lane >> 5) * 16;
    const int c0 = lane & 31;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (c0 == r) p[r] = v[r];
  }
  __device__ __forceinline__ void get_partner(int area, float (&v)[16]) const {
    const float* p = base + area * 256 + (wave ^ 4) * 32 + (lane >> 5) * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(p + i * 4);
      v[i * 4 + 0] = t[0]; v[i * 4 + 1] = t[1]; v[i * 4 + 2] = t[2]; v[i * 4 + 3] = t[3];
    }
  }
};

// s_waitcnt vmcnt(N) alone (expcnt / lgkmcnt fields left at their maxima)
template <int N>
__device__ __forceinline__ void wait_vm_count() {
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}

// NTW = accumulators per wave: 4 -> 256-column blocks (the common case), 2 -> 128-column blocks
// (coordinate gate, atom head), whose stage is 128 + 128 rows and whose waves issue two DMAs per chunk.
// SLABS = 32-row slabs per workgroup (2 * SLABS waves): 4 -> 128 rows, the form for launches that fill the chip.
// A launch of few rows (M / 128 well below the number of CUs) puts its waves two to a SIMD on a fraction of
// the CUs while the rest idle, and a wave's 32 x 128 x K tile is a fixed chain of K / 2 x 4 MFMAs of 64 cycles:
// with SLABS = 2 or 1 the same waves spread over four or eight times as many CUs (the weight chunk is then
// copied once per 64 or 32 rows instead of once per 128, from L2).  A row's products and sums are the same
// instructions in the same order for every SLABS, so the result does not depend on it.
template <class AP, class EPI, int NTW = 4, int SLABS = 4>
__global__ __launch_bounds__(128 * SLABS, 2) void gemm_cs_kernel(GemmCore g, AP ap, EPI epi) {
  constexpr int A_FL = 32 * SLABS * G2_ROWF;
  constexpr int ST = (32 * SLABS + 64 * NTW) * G2_ROWF;
  constexpr int WP = 2 * NTW / SLABS;   // 16-row W pieces per wave and chunk (4 * NTW pieces over 2 * SLABS waves)
  __shared__ __attribute__((aligned(1024))) float smem3[3 * ST + 4 * 256];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slab = wave % SLABS, half = wave / SLABS;
  const int by = blockIdx.y;
  const int row_blk = blockIdx.x * (32 * SLABS);
  const float* __restrict__ W = g.W[by];

  f32x16 acc[NTW];
#pragma unroll
  for (int ct = 0; ct < NTW; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;

  // DMA roles (see gemm2.h): one A piece (16 rows) and WP W pieces per wave and chunk
  const int pr = lane >> 2, ps = lane & 3;
  const int lslot = ps ^ ((pr >> 2) & 3);
  int a_row = row_blk + wave * 16 + pr;
  a_row = a_row < g.M ? a_row : g.M - 1;
  const float* w_src[WP];
#pragma unroll
  for (int i = 0; i < WP; ++i) {
    int n = (wave * WP + i) * 16 + pr;
    n = n < g.n_valid ? n : g.n_valid - 1;
    w_src[i] = W + (size_t)n * g.ldw + lslot * 4;
  }
  auto issue = [&](int kc, int buf) {
    float* sA = smem3 + buf * ST;
    float* sW = sA + A_FL;
    const int k0 = kc * G2_BK;
    lds_dma16(ap.ptr(a_row, k0 + lslot * 4), sA + wave * 256);
#pragma unroll
    for (int i = 0; i < WP; ++i) lds_dma16(w_src[i] + k0, sW + (wave * WP + i) * 256);
  };

  const int nk = g.K / G2_BK;
  const int fi = lane & 31, hk = lane >> 5;
  const int swz = (fi >> 2) & 3;
  const int off0 = ((0 * 2 + hk) ^ swz) * 4, off1 = ((1 * 2 + hk) ^ swz) * 4;
  issue(0, 0);
  if (nk > 1) issue(1, 1);
  int cur = 0;
  for (int kc = 0; kc < nk; ++kc) {
    // own DMAs of chunk kc done (the newest chunk may stay in flight), then join: every
    // wave's part of chunk kc is in LDS and nobody still reads the stage refilled below
    if (kc + 1 < nk) wait_vm_count<1 + WP>();
    else wait_vm_count<0>();
    __builtin_amdgcn_s_barrier();
    if (kc + 2 < nk) issue(kc + 2, cur == 0 ? 2 : cur - 1);
    const float* sA = smem3 + cur * ST;
    const float* sW = sA + A_FL + (half * 32 * NTW + fi) * G2_ROWF;
    const float* arow = sA + (slab * 32 + fi) * G2_ROWF;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(arow + off0);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(arow + off1);
    f32x4 b[2][2];
    b[0][0] = *reinterpret_cast<const f32x4*>(sW + off0);
    b[0][1] = *reinterpret_cast<const f32x4*>(sW + off1);
#pragma unroll
    for (int ct = 0; ct < NTW; ++ct) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b[ct & 1][0][j], acc[ct], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (ct + 1 < NTW) {
        b[(ct + 1) & 1][0] = *reinterpret_cast<const f32x4*>(sW + (ct + 1) * 32 * G2_ROWF + off0);
        b[(ct + 1) & 1][1] = *reinterpret_cast<const f32x4*>(sW + (ct + 1) * 32 * G2_ROWF + off1);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b[ct & 1][1][j], acc[ct], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    cur = cur == 2 ? 0 : cur + 1;
  }
  HalfMailbox mb{smem3 + 3 * ST, slab + 4 * half, lane};   // mailbox slots are numbered as in the 8-wave form
  epi.cs(acc, row_blk + slab * 32, by, half, lane, mb);
}

// 32-row form for launches of very few rows (at most one workgroup per CU): FOUR waves per 32-row slab, wave =
// column half x column quarter, i.e. a 32 x 64 tile of two accumulators, so the chain of MFMAs a wave runs is half
// that of the forms above and each wave has a SIMD to itself.  The epilogues want a wave to hold 128 columns of its
// rows: after the K loop the second quarter of each half hands its two accumulators to the first through LDS and
// ends; the first runs EPI::cs unchanged (a workgroup barrier inside an epilogue then joins the surviving waves
// only).  An accumulator still receives the same MFMAs in the same order, so the results are bit for bit those of
// gemm_cs_kernel.  Both waves of a quarter pair copy the slab's two A pieces (identical bytes to the same LDS
// address) so that every wave issues five DMAs per chunk and one wait count serves all.
template <class AP, class EPI>
__global__ __launch_bounds__(256, 2) void gemm_cs_quarter_kernel(GemmCore g, AP ap, EPI epi) {
  constexpr int A_FL = 32 * G2_ROWF;
  constexpr int ST = (32 + 256) * G2_ROWF;
  __shared__ __attribute__((aligned(1024))) float smemq[3 * ST + 4 * 256];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 1, quarter = wave & 1;
  const int by = blockIdx.y;
  const int row_blk = blockIdx.x * 32;
  const float* __restrict__ W = g.W[by];

  f32x16 acc[4];   // the wave's own tiles are acc[0], acc[1] until the hand-over
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;

  const int pr = lane >> 2, ps = lane & 3;
  const int lslot = ps ^ ((pr >> 2) & 3);
  int a_row = row_blk + quarter * 16 + pr;
  a_row = a_row < g.M ? a_row : g.M - 1;
  const float* w_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int n = (wave * 4 + i) * 16 + pr;
    n = n < g.n_valid ? n : g.n_valid - 1;
    w_src[i] = W + (size_t)n * g.ldw + lslot * 4;
  }
  auto issue = [&](int kc, int buf) {
    float* sA = smemq + buf * ST;
    float* sW = sA + A_FL;
    const int k0 = kc * G2_BK;
    lds_dma16(ap.ptr(a_row, k0 + lslot * 4), sA + quarter * 256);
#pragma unroll
    for (int i = 0; i < 4; ++i) lds_dma16(w_src[i] + k0, sW + (wave * 4 + i) * 256);
  };

  const int nk = g.K / G2_BK;
  const int fi = lane & 31, hk = lane >> 5;
  const int swz = (fi >> 2) & 3;
  const int off0 = ((0 * 2 + hk) ^ swz) * 4, off1 = ((1 * 2 + hk) ^ swz) * 4;
  issue(0, 0);
  if (nk > 1) issue(1, 1);
  int cur = 0;
  for (int kc = 0; kc < nk; ++kc) {
    if (kc + 1 < nk) wait_vm_count<5>();
    else wait_vm_count<0>();
    __builtin_amdgcn_s_barrier();
    if (kc + 2 < nk) issue(kc + 2, cur == 0 ? 2 : cur - 1);
    const float* sA = smemq + cur * ST;
    const float* sW = sA + A_FL + (half * 128 + quarter * 64 + fi) * G2_ROWF;
    const float* arow = sA + fi * G2_ROWF;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(arow + off0);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(arow + off1);
    f32x4 b[2][2];
    b[0][0] = *reinterpret_cast<const f32x4*>(sW + off0);
    b[0][1] = *reinterpret_cast<const f32x4*>(sW + off1);
    b[1][0] = *reinterpret_cast<const f32x4*>(sW + 32 * G2_ROWF + off0);
    b[1][1] = *reinterpret_cast<const f32x4*>(sW + 32 * G2_ROWF + off1);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b[ct][0][j], acc[ct], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b[ct][1][j], acc[ct], 0, 0, 0);
    }
    cur = cur == 2 ? 0 : cur + 1;
  }
  // hand-over through the stage area ([half][tile 2][register 16][lane 64] floats = 16 KiB), free once every wave
  // has left the K loop
  __syncthreads();
  float* hand = smemq + half * (2 * 16 * 64);
  if (quarter == 1) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) hand[(ct * 16 + r) * 64 + lane] = acc[ct][r];
  }
  __syncthreads();
  if (quarter == 1) return;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[2 + ct][r] = hand[(ct * 16 + r) * 64 + lane];
  HalfMailbox mb{smemq + 3 * ST, 4 * half, lane};   // slots of slab 0 in the 8-wave numbering
  epi.cs(acc, row_blk, by, half, lane, mb);
}
