// Row-slab fp32 MFMA GEMM, LDS-DMA staged (second-generation main loop of gemm.h).
//
// Same tile and accumulator ownership as gemm_rows_kernel (128 rows x 32*NT columns per
// workgroup, wave w owns rows [32w, 32w+32) and all columns, v_mfma_f32_32x32x2_f32), so
// every row-wise epilogue is shared.  What differs is how operands reach LDS:
//   * K is consumed in chunks of 16 (64-B rows).  Both tiles are written by
//     global_load_lds_dwordx4 (no staging registers): one wave-instruction lands 1 KiB =
//     16 rows x 64 B, lane l -> row l>>2, 16-B slot l&3.
//   * LDS rows are unpadded (the DMA writes linearly); bank conflicts of the ds_read_b128
//     fragment reads are removed by an XOR swizzle applied on the SOURCE side: the lane
//     that fills physical slot p of row r fetches logical slot p ^ ((r>>2)&3), and a reader
//     of logical slot q of row r reads physical slot q ^ ((r>>2)&3).
//   * Two LDS buffers: the DMA of chunk k+1 is issued right after the barrier that opens
//     chunk k and has the whole chunk (64 MFMAs per wave) to land; one barrier per chunk.
//   * B fragments are read one column tile ahead of the MFMAs that consume them.
// Rows beyond M / n_valid are CLAMPED to the last valid row when loading (their products
// only reach outputs that the epilogues never store).  K must be a multiple of 16.
#pragma once
#include "gemm.h"

#define G2_BK 16
#define G2_ROWF 16  // floats per LDS row

// A-operand address providers: ptr(row, k) for a VALID row, k multiple of 4.
struct APtrPlain {
  const float* A; int lda;
  __device__ __forceinline__ const float* ptr(int row, int k) const { return A + (size_t)row * lda + k; }
};
struct APtrConcat {  // [A | A2] along k, split at ksplit (multiple of 16)
  const float* A; const float* A2; int lda; int lda2; int ksplit;
  __device__ __forceinline__ const float* ptr(int row, int k) const {
    return k < ksplit ? A + (size_t)row * lda + k : A2 + (size_t)row * lda2 + (k - ksplit);
  }
};

__device__ __forceinline__ void lds_dma16(const float* g, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int NT, class AP, class EPI>
__global__ __launch_bounds__(256, 2) void gemm_rows2_kernel(GemmCore g, AP ap, EPI epi) {
  constexpr int A_FL = G_BM * G2_ROWF;        // floats per A buffer (8 KiB)
  constexpr int W_FL = NT * 32 * G2_ROWF;     // floats per W buffer
  __shared__ __attribute__((aligned(1024))) float smem[2 * (A_FL + W_FL)];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int by = blockIdx.y;
  const int row_blk = blockIdx.x * G_BM;
  const float* __restrict__ W = g.W[by];

  f32x16 acc[NT];
#pragma unroll
  for (int ct = 0; ct < NT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;

  // DMA lane roles: within one 1-KiB piece lane l fills row (l>>2), physical slot (l&3)
  const int pr = lane >> 2, ps = lane & 3;
  const int lslot = ps ^ ((pr >> 2) & 3);      // logical slot this lane fetches (16 rows per piece: r>>2 = pr>>2)
  // A tile: 8 pieces of 16 rows -> 2 per wave; W tile: NT*2 pieces -> NT/2 per wave
  int a_row[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int row = row_blk + (wave * 2 + i) * 16 + pr;
    a_row[i] = row < g.M ? row : g.M - 1;
  }
  const float* w_src[NT / 2];
#pragma unroll
  for (int i = 0; i < NT / 2; ++i) {
    int n = (wave * (NT / 2) + i) * 16 + pr;
    n = n < g.n_valid ? n : g.n_valid - 1;
    w_src[i] = W + (size_t)n * g.ldw + lslot * 4;
  }

  auto issue = [&](int kc, int buf) {
    float* sA = smem + buf * (A_FL + W_FL);
    float* sW = sA + A_FL;
    const int k0 = kc * G2_BK;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      lds_dma16(ap.ptr(a_row[i], k0 + lslot * 4), sA + (wave * 2 + i) * 256);
#pragma unroll
    for (int i = 0; i < NT / 2; ++i) lds_dma16(w_src[i] + k0, sW + (wave * (NT / 2) + i) * 256);
  };

  const int nk = g.K / G2_BK;
  const int fi = lane & 31, hk = lane >> 5;
  const int swz = (fi >> 2) & 3;
  // physical 16-B slot of logical slot (kk*2 + hk) in this lane's row, for kk = 0, 1
  const int off0 = ((0 * 2 + hk) ^ swz) * 4, off1 = ((1 * 2 + hk) ^ swz) * 4;
  issue(0, 0);
  for (int kc = 0; kc < nk; ++kc) {
    __syncthreads();  // waits this wave's DMA (vmcnt) and joins: chunk kc landed, buffer (kc+1)&1 free
    if (kc + 1 < nk) issue(kc + 1, (kc + 1) & 1);
    const float* sA = smem + (kc & 1) * (A_FL + W_FL);
    const float* sW = sA + A_FL;
    const float* arow = sA + (wave * 32 + fi) * G2_ROWF;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(arow + off0);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(arow + off1);
    f32x4 b[2][2];
    b[0][0] = *reinterpret_cast<const f32x4*>(sW + fi * G2_ROWF + off0);
    b[0][1] = *reinterpret_cast<const f32x4*>(sW + fi * G2_ROWF + off1);
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
      // first half of this tile's MFMAs: the only outstanding LDS reads are this tile's
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b[ct & 1][0][j], acc[ct], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (ct + 1 < NT) {  // next tile's fragments, 4 MFMAs (256 cycles) ahead of their first use
        b[(ct + 1) & 1][0] = *reinterpret_cast<const f32x4*>(sW + ((ct + 1) * 32 + fi) * G2_ROWF + off0);
        b[(ct + 1) & 1][1] = *reinterpret_cast<const f32x4*>(sW + ((ct + 1) * 32 + fi) * G2_ROWF + off1);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b[ct & 1][1][j], acc[ct], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  epi(acc, row_blk + wave * 32, by, lane);
}
