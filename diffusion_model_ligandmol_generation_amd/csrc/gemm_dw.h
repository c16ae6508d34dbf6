// Weight-gradient GEMM on the LDS-DMA pipeline:  dW[n, k] += sum_m dY[m, n] X[m, k]
// (optionally db[n] += sum_m dY[m, n] from the same pass).
//
// Both operands have the reduction index m as their ROW index, so a 16-deep chunk of either is
// 16 full rows - exactly what global_load_lds moves.  A workgroup (8 waves) owns a 128 (n) x
// 256 (k) tile of dW over one slice of m (blockIdx.z): wave = 32-row n slab x 128-column k half,
// four v_mfma_f32_32x32x2_f32 accumulators, three stage buffers, counted vmcnt waits - the structure
// of gemm_cs_kernel.  In LDS a stage is [16 m][128 n] + [16 m][256 k]; an MFMA operand needs, for
// lane (i, h), element [m = 2s + h][i]: consecutive lanes read consecutive floats, and odd m rows are
// stored rotated by 32 floats (applied on the DMA source address) so that the two lane halves,
// which read rows m and m + 1, hit different banks.  Slices add their tiles with float atomics
// into the pre-zeroed gradient.  Rows of the last, partial chunk beyond the slice read as zero
// through the A operand (the addresses are clamped, the value is masked).
#pragma once
#include "gemm2.h"

// NTW = accumulators per wave: 4 -> 256 k columns per workgroup, 2 -> 128 (for small gradients, where
// twice the tiles means half the slices and half the atomic traffic on each address).

struct GemmDwArgs {
  const float* dY; int lddy; int nout;    // [rows, nout]
  const float* X; int ldx; int kin;       // [rows, kin]   (kin a multiple of the 64 NTW columns of a grid.y block)
  int rows; int m_split;                  // m range per blockIdx.z (multiple of 16)
  float* dW; int lddw;                    // [nout, kin], pre-zeroed, accumulated atomically
  float* db;                              // nullable: [nout] bias gradient
};

template <int NTW>
__global__ __launch_bounds__(512, 2) void gemm_dw_kernel(GemmDwArgs g) {
  constexpr int KC = 64 * NTW;                       // k columns per workgroup
  constexpr int ST = 16 * 128 + 16 * KC, A_FL = 16 * 128;
  __shared__ __attribute__((aligned(1024))) float smem[3 * ST];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slab = wave & 3, half = wave >> 2;
  const int n_blk = blockIdx.x * 128, k_blk = blockIdx.y * KC;
  const int m_begin = blockIdx.z * g.m_split;
  const int m_end = min(g.rows, m_begin + g.m_split);
  if (m_begin >= m_end) return;

  f32x16 acc[NTW];
#pragma unroll
  for (int ct = 0; ct < NTW; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;

  // DMA roles.  A stage: 16 rows x 512 B = 8 pieces of 2 rows (wave w -> rows 2w, 2w+1; lane l ->
  // row 2w + (l >> 5), physical 16-B slot l & 31).  B stage: 16 rows x 1 KiB = 16 pieces of one row
  // (wave w -> rows 2w, 2w+1; lane l -> physical slot l).  Odd rows are rotated by 8 slots (32 floats).
  const int a_m = wave * 2 + (lane >> 5);
  const int a_slot = ((lane & 31) - 8 * (a_m & 1)) & 31;        // logical slot fetched into physical slot l & 31
  int a_col = n_blk + a_slot * 4;
  a_col = a_col + 3 < g.nout ? a_col : (g.nout >= 4 ? ((g.nout - 4) & ~3) : 0);   // clamp inside the row (masked later)
  // B stage, NTW = 4: 16 rows x 1 KiB = 16 one-row pieces, two per wave (rows 2w, 2w + 1; lane l -> slot l);
  //          NTW = 2: 16 rows x 512 B = 8 two-row pieces, one per wave (same shape as the A stage).
  int b_slot[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) b_slot[i] = (lane - 8 * ((wave * 2 + i) & 1)) & 63;
  const int b2_slot = ((lane & 31) - 8 * (a_m & 1)) & 31;
  auto issue = [&](int chunk, int buf) {
    float* sA = smem + buf * ST;
    float* sB = sA + A_FL;
    const int m0 = m_begin + chunk * 16;
    int ma = m0 + a_m;
    ma = ma < g.rows ? ma : g.rows - 1;
    lds_dma16(g.dY + (size_t)ma * g.lddy + a_col, sA + wave * 256);
    if (NTW == 4) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int mb = m0 + wave * 2 + i;
        mb = mb < g.rows ? mb : g.rows - 1;
        lds_dma16(g.X + (size_t)mb * g.ldx + k_blk + b_slot[i] * 4, sB + (wave * 2 + i) * 256);
      }
    } else {
      lds_dma16(g.X + (size_t)ma * g.ldx + k_blk + b2_slot * 4, sB + wave * 256);
    }
  };

  const int nchunks = (m_end - m_begin + 15) / 16;
  const int fi = lane & 31, hk = lane >> 5;
  // column of this lane inside a stage row, for even (h = 0 reads rows 2s) and odd rows
  const int a_off = (slab * 32 + fi + 32 * hk) & 127;             // rotated by 32 on odd rows (m = 2s + 1)
  float bias_part = 0.f;
  issue(0, 0);
  if (nchunks > 1) issue(1, 1);
  int cur = 0;
  for (int c = 0; c < nchunks; ++c) {
    if (c + 1 < nchunks) __builtin_amdgcn_s_waitcnt(0x0F70 | (1 + NTW / 2));   // vmcnt(DMAs per chunk)
    else __builtin_amdgcn_s_waitcnt(0x0F70);                                   // vmcnt(0)
    __builtin_amdgcn_s_barrier();
    if (c + 2 < nchunks) issue(c + 2, cur == 0 ? 2 : cur - 1);
    const float* sA = smem + cur * ST;
    const float* sB = sA + A_FL;
    float a[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) a[s] = sA[(2 * s + hk) * 128 + a_off];
    const int m_left = m_end - (m_begin + c * 16);   // valid rows in this chunk
    if (m_left < 16) {
#pragma unroll
      for (int s = 0; s < 8; ++s) a[s] = (2 * s + hk) < m_left ? a[s] : 0.f;
    }
    if (g.db != nullptr) {
#pragma unroll
      for (int s = 0; s < 8; ++s) bias_part += a[s];
    }
#pragma unroll
    for (int ct = 0; ct < NTW; ++ct) {
      const int b_off = (half * 32 * NTW + ct * 32 + fi + 32 * hk) & (KC - 1);
#pragma unroll
      for (int s = 0; s < 8; ++s)
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], sB[(2 * s + hk) * KC + b_off], acc[ct], 0, 0, 0);
    }
    cur = cur == 2 ? 0 : cur + 1;
  }
  // epilogue: accumulator row = n (slab row), column = k
  const int c0 = lane & 31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = n_blk + slab * 32 + G_ROW(r, lane);
    if (n < g.nout) {
#pragma unroll
      for (int ct = 0; ct < NTW; ++ct) {
        const int k = k_blk + half * 32 * NTW + ct * 32 + c0;
        if (k < g.kin) atomicAdd(g.dW + (size_t)n * g.lddw + k, acc[ct][r]);
      }
    }
  }
  if (g.db != nullptr && blockIdx.y == 0 && half == 0) {
    bias_part += __shfl_xor(bias_part, 32);
    const int n = n_blk + slab * 32 + fi;
    if (hk == 0 && n < g.nout) atomicAdd(g.db + n, bias_part);
  }
}
