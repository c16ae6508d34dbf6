// Row-slab fp32 MFMA GEMM for gfx950:  C[M, NB*BN] = epilogue( A[M,K] * W[n,K]^T ).
//
// One workgroup (4 waves) computes a 128-row x (32*NT)-column block; wave w owns rows
// [32w, 32w+32) and ALL columns of the block as NT accumulators of
// v_mfma_f32_32x32x2_f32.  Because a wave holds complete rows, row-wise epilogues
// (LayerNorm, gated residual + modulated LayerNorm, dot-product heads) run on the
// accumulators with half-wave shuffles only - no second kernel and no LDS round trip.
//
// Accumulator map of the 32x32 tile (dtype independent on gfx950):
//   col = lane & 31,  row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5),  reg in [0,16).
// Operand map of v_mfma_f32_32x32x2_f32: lane l feeds A[row = l & 31][k = l >> 5] and
// B[k = l >> 5][col = l & 31].  Both operands are staged in LDS with k contiguous and
// each lane reads 4 consecutive k (ds_read_b128) for 4 successive MFMAs, so the k order
// inside a 32-deep chunk is permuted identically for A and B (sum order only).
#pragma once
#include "common.h"

#define G_BM 128
#define G_BK 32
#define G_LDS 36  // floats per LDS row: 32 + 4 pad -> 144-B stride, conflict-free b128 reads

#define G_ROW(r, lane) (((r) & 3) + 8 * ((r) >> 2) + 4 * ((lane) >> 5))

struct GemmCore {
  int M;        // rows of A / C
  int K;        // reduction length (zero-extended to a multiple of 32)
  int n_valid;  // valid rows of W in one column block (<= 32*NT); the rest read as zero
  int ldw;      // leading dimension of W
  const float* W[8];  // W[by]: first row of column block by, k contiguous
};

// ---------------------------------------------------------------- A-operand loaders
// load(m, k): 4 consecutive k of row m; zero outside [0,M) x [0,K).
struct ALoadPlain {  // lda % 4 == 0, K % 4 == 0, 16-B aligned base
  const float* A; int lda; int M; int K;
  __device__ __forceinline__ f32x4 load(int m, int k) const {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (m < M && k < K) v = *reinterpret_cast<const f32x4*>(A + (size_t)m * lda + k);
    return v;
  }
};
struct ALoadScalar {  // any lda / K (e.g. x[N,119])
  const float* A; int lda; int M; int K;
  __device__ __forceinline__ f32x4 load(int m, int k) const {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (m < M) {
      const float* p = A + (size_t)m * lda + k;
      if (k + 0 < K) v[0] = p[0];
      if (k + 1 < K) v[1] = p[1];
      if (k + 2 < K) v[2] = p[2];
      if (k + 3 < K) v[3] = p[3];
    }
    return v;
  }
};
struct ALoadConcat {  // [A | A2] along k, split at ksplit (multiple of 4)
  const float* A; const float* A2; int lda; int lda2; int ksplit; int M; int K;
  __device__ __forceinline__ f32x4 load(int m, int k) const {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (m < M && k < K) {
      v = (k < ksplit) ? *reinterpret_cast<const f32x4*>(A + (size_t)m * lda + k)
                       : *reinterpret_cast<const f32x4*>(A2 + (size_t)m * lda2 + (k - ksplit));
    }
    return v;
  }
};
// A given TRANSPOSED in memory: A_t[k][m] (m contiguous).  Used for dW = dY^T X where
// the reduction index is the row index of both operands.
struct ALoadTrans {
  const float* At; int ld; int M; int K;  // element (m,k) at At[k*ld + m]
  __device__ __forceinline__ f32x4 load(int m, int k) const {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (m < M) {
      if (k + 0 < K) v[0] = At[(size_t)(k + 0) * ld + m];
      if (k + 1 < K) v[1] = At[(size_t)(k + 1) * ld + m];
      if (k + 2 < K) v[2] = At[(size_t)(k + 2) * ld + m];
      if (k + 3 < K) v[3] = At[(size_t)(k + 3) * ld + m];
    }
    return v;
  }
};

// W loader modes
enum { W_ALIGNED = 0, W_SCALAR = 1, W_TRANS = 2 };
template <int MODE>
__device__ __forceinline__ f32x4 gemm_wload(const float* __restrict__ W, int ldw, int n, int k,
                                            int n_valid, int K) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (n < n_valid) {
    if (MODE == W_ALIGNED) {
      if (k < K) v = *reinterpret_cast<const f32x4*>(W + (size_t)n * ldw + k);
    } else if (MODE == W_SCALAR) {
      const float* p = W + (size_t)n * ldw + k;
      if (k + 0 < K) v[0] = p[0];
      if (k + 1 < K) v[1] = p[1];
      if (k + 2 < K) v[2] = p[2];
      if (k + 3 < K) v[3] = p[3];
    } else {  // W stored [K][n] (n contiguous): element (n,k) at W[k*ldw + n]
      if (k + 0 < K) v[0] = W[(size_t)(k + 0) * ldw + n];
      if (k + 1 < K) v[1] = W[(size_t)(k + 1) * ldw + n];
      if (k + 2 < K) v[2] = W[(size_t)(k + 2) * ldw + n];
      if (k + 3 < K) v[3] = W[(size_t)(k + 3) * ldw + n];
    }
  }
  return v;
}

template <int NT, int WMODE, class AL, class EPI>
__global__ __launch_bounds__(256, 2) void gemm_rows_kernel(GemmCore g, AL al, EPI epi) {
  __shared__ __attribute__((aligned(16))) float sA[G_BM * G_LDS];
  __shared__ __attribute__((aligned(16))) float sW[NT * 32 * G_LDS];
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

  f32x4 ra[4], rw[NT];
  const int st_r = tid >> 3;        // staging row (0..31) within a 32-row pass
  const int st_c = (tid & 7) * 4;   // staging k offset
  const int nk = (g.K + G_BK - 1) / G_BK;

  auto gload = [&](int kc) {
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = al.load(row_blk + st_r + i * 32, kc * G_BK + st_c);
#pragma unroll
    for (int i = 0; i < NT; ++i)
      rw[i] = gemm_wload<WMODE>(W, g.ldw, st_r + i * 32, kc * G_BK + st_c, g.n_valid, g.K);
  };

  gload(0);
  const int fi = lane & 31, hk = lane >> 5;
  for (int kc = 0; kc < nk; ++kc) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<f32x4*>(&sA[(st_r + i * 32) * G_LDS + st_c]) = ra[i];
#pragma unroll
    for (int i = 0; i < NT; ++i)
      *reinterpret_cast<f32x4*>(&sW[(st_r + i * 32) * G_LDS + st_c]) = rw[i];
    __syncthreads();
    if (kc + 1 < nk) gload(kc + 1);  // in flight while this chunk is multiplied

    f32x4 a[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      a[kk] = *reinterpret_cast<const f32x4*>(&sA[(wave * 32 + fi) * G_LDS + kk * 8 + hk * 4]);
    // B fragments are read one column tile ahead of the MFMAs that consume them
    f32x4 b[2][4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      b[0][kk] = *reinterpret_cast<const f32x4*>(&sW[fi * G_LDS + kk * 8 + hk * 4]);
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
      if (ct + 1 < NT) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          b[(ct + 1) & 1][kk] =
              *reinterpret_cast<const f32x4*>(&sW[((ct + 1) * 32 + fi) * G_LDS + kk * 8 + hk * 4]);
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][j], b[ct & 1][kk][j], acc[ct], 0, 0, 0);
    }
    __syncthreads();
  }
  epi(acc, row_blk + wave * 32, by, lane);
}

// ------------------------------------------------------------------------ epilogues
// Every epilogue has two entry points: operator() for kernels whose wave holds all columns
// of its rows (NT accumulators), and cs() for the column-split kernel of gemm3.h, where the
// wave holds columns [128 half, 128 half + 128) as four accumulators and per-row partial
// results cross to the partner wave through the HalfMailbox.
struct HalfMailbox;
enum { ACT_NONE = 0, ACT_GELU = 1, ACT_SILU = 2 };
template <int ACT>
__device__ __forceinline__ float apply_act(float v) {
  if (ACT == ACT_GELU) return gelu_exact(v);
  if (ACT == ACT_SILU) return silu(v);
  return v;
}

// C[row, by*col_block + col] = act(acc + bias);  optional C2 = silu(C);
// optional ACCUM: C += value (used by split reductions that run serially).
// DROP (column-split form only): the activation's output is multiplied by the dropout mask of
// site `drop` before it is stored (model1.py:270: Dropout after the FFN's GELU); pre_out keeps u.
template <int NT, int ACT, bool DUAL_SILU, bool DROP = false>
struct EpiBiasAct {
  int M; int n_valid; int col_block;
  const float* bias[8];
  float* C; int ldc;
  float* C2;
  float alpha;     // scales acc before bias (1.0f normally)
  float* pre_out;  // nullable: u = alpha*acc + bias before the activation (kept for backward)
  DropSite drop;   // used with DROP only
  __device__ __forceinline__ void operator()(f32x16 (&acc)[NT], int row0, int by, int lane) const {
    const int c0 = lane & 31;
    const float* b = bias[by];
    float bv[NT];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
      const int col = ct * 32 + c0;
      bv[ct] = (b != nullptr && col < n_valid) ? b[col] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + G_ROW(r, lane);
      if (row < M) {
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
          const int col = ct * 32 + c0;
          if (col < n_valid) {
            const float u = acc[ct][r] * alpha + bv[ct];
            const float v = apply_act<ACT>(u);
            const size_t o = (size_t)row * ldc + (size_t)by * col_block + col;
            C[o] = v;
            if (pre_out != nullptr) pre_out[o] = u;
            if (DUAL_SILU) C2[o] = silu(v);
          }
        }
      }
    }
  }
  // Column-split form (NW accumulators per wave: columns [32 NW half, 32 NW (half + 1)) of the block).
  // The optional outputs are a template switch so that the row loop has no per-element branch, and
  // GELU runs two columns at a time on packed fp32 instructions.
  template <bool PRE, int NW, class MB>
  __device__ __forceinline__ void cs_body(f32x16 (&acc)[NW], int row0, int by, int half, int lane, const MB&) const {
    const int c0 = half * 32 * NW + (lane & 31);
    const float* b = bias[by];
    float bv[NW];
    bool ok[NW];
#pragma unroll
    for (int ct = 0; ct < NW; ++ct) {
      ok[ct] = c0 + ct * 32 < n_valid;
      bv[ct] = (b != nullptr && ok[ct]) ? b[c0 + ct * 32] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + G_ROW(r, lane);
      if (row < M) {
        const size_t o0 = (size_t)row * ldc + (size_t)by * col_block + c0;
        float u[NW], v[NW];
#pragma unroll
        for (int ct = 0; ct < NW; ++ct) u[ct] = acc[ct][r] * alpha + bv[ct];
        if (ACT == ACT_GELU) {
#pragma unroll
          for (int ct = 0; ct < NW; ct += 2) {
            const f32x2 g0 = gelu_exact2(f32x2{u[ct], u[ct + 1]});
            v[ct] = g0[0]; v[ct + 1] = g0[1];
          }
        } else {
#pragma unroll
          for (int ct = 0; ct < NW; ++ct) v[ct] = apply_act<ACT>(u[ct]);
        }
        if (DROP) {
          static_assert(!DROP || NW == 4, "dropout epilogue: 256-column blocks only");
          const f32x4 ms = drop_scale4(drop, drop_row_quad(drop, row, by * col_block + c0, ldc));
#pragma unroll
          for (int ct = 0; ct < NW; ++ct) v[ct] *= ms[ct & 3];
        }
#pragma unroll
        for (int ct = 0; ct < NW; ++ct) {
          if (NW == 4 || ok[ct]) {   // 256-column blocks are always full
            C[o0 + ct * 32] = v[ct];
            if (PRE) pre_out[o0 + ct * 32] = u[ct];
            if (DUAL_SILU) C2[o0 + ct * 32] = silu(v[ct]);
          }
        }
      }
    }
  }
  template <int NW, class MB>
  __device__ __forceinline__ void cs(f32x16 (&acc)[NW], int row0, int by, int half, int lane, const MB& mb) const {
    if (pre_out != nullptr) cs_body<true, NW>(acc, row0, by, half, lane, mb);
    else cs_body<false, NW>(acc, row0, by, half, lane, mb);
  }
};

// y = post(LN(pre(acc + bias)) * gamma + beta) [+ resid]; each column block is one
// 256-wide LayerNorm group (NT = 8).  pre_out, if set, also receives u = acc + bias, from
// which the backward pass recomputes the activation and the row statistics.
template <int PRE_ACT, int POST_ACT, bool RESID>
struct EpiLN {
  int M;
  const float* bias[8]; const float* gamma[8]; const float* beta[8];
  float eps;
  float* C; int ldc;
  const float* resid; int ldr;
  float* pre_out;    // nullable: [M, ldc] u = acc + bias
  __device__ __forceinline__ void operator()(f32x16 (&acc)[8], int row0, int by, int lane) const {
    const int c0 = lane & 31;
    float bv[8], gv[8], be[8];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
      const int col = ct * 32 + c0;
      bv[ct] = bias[by][col]; gv[ct] = gamma[by][col]; be[ct] = beta[by][col];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + G_ROW(r, lane);
      float v[8], s = 0.f;
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) { v[ct] = apply_act<PRE_ACT>(acc[ct][r] + bv[ct]); s += v[ct]; }
      const float mean = half_wave_sum(s) * (1.0f / 256.0f);
      float q = 0.f;
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) { const float d = v[ct] - mean; q += d * d; }
      const float rstd = 1.0f / sqrtf(half_wave_sum(q) * (1.0f / 256.0f) + eps);
      if (row < M) {
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
          const int col = ct * 32 + c0;
          const size_t o = (size_t)row * ldc + (size_t)by * 256 + col;
          float y = apply_act<POST_ACT>((v[ct] - mean) * rstd * gv[ct] + be[ct]);
          if (RESID) y += resid[(size_t)row * ldr + (size_t)by * 256 + col];
          C[o] = y;
          if (pre_out != nullptr) pre_out[o] = acc[ct][r] + bv[ct];
        }
      }
    }
  }
  // Loads that do not depend on the accumulators (the residual rows) are issued first, in one
  // batch: the compiler may not move a load above an earlier store through a possibly aliasing
  // pointer (C may BE resid), so loads left inside the row loop would each wait out a full
  // memory latency behind the previous row's stores.
  template <bool SAVE, class MB>
  __device__ __forceinline__ void cs_body(f32x16 (&acc_t)[4], int row0, int by, int half, int lane, const MB& mb) const {
    float acc[4][16];   // plain scalars: element updates on the 16-register MFMA tuples would copy whole tuples
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][r] = acc_t[ct][r];
    const int c0 = lane & 31;
    const int cb = by * 256 + half * 128;
    float bv[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) bv[ct] = bias[by][half * 128 + ct * 32 + c0];
    float mean[16], m2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + G_ROW(r, lane);
      float s = 0.f;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const float u = acc[ct][r] + bv[ct];
        if (SAVE && row < M) pre_out[(size_t)row * ldc + cb + ct * 32 + c0] = u;
        acc[ct][r] = apply_act<PRE_ACT>(u);
        s += acc[ct][r];
      }
      mean[r] = half_wave_sum(s) * (1.0f / 128.0f);
      float q = 0.f;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) { const float d = acc[ct][r] - mean[r]; q += d * d; }
      m2[r] = half_wave_sum(q);
      __builtin_amdgcn_sched_barrier(0);
    }
    mb.put(0, mean); mb.put(1, m2);
    // residual rows: all loads go out before the barrier and before the first store
    float rs[RESID ? 4 : 1][16];
    if (RESID) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + G_ROW(r, lane);
        const int rr = row < M ? row : M - 1;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) rs[ct][r] = resid[(size_t)rr * ldr + cb + ct * 32 + c0];
      }
    }
    __syncthreads();
    {
      float pmean[16], pm2[16];
      mb.get_partner(0, pmean); mb.get_partner(1, pm2);
#pragma unroll
      for (int r = 0; r < 16; ++r) {   // mean[] <- row mean, m2[] <- 1/std
        const float dm = mean[r] - pmean[r];
        const float var = (m2[r] + pm2[r] + 64.0f * dm * dm) * (1.0f / 256.0f);
        mean[r] = 0.5f * (mean[r] + pmean[r]);
        m2[r] = 1.0f / sqrtf(var + eps);
      }
    }
    float gv[4], be[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) { gv[ct] = gamma[by][half * 128 + ct * 32 + c0]; be[ct] = beta[by][half * 128 + ct * 32 + c0]; }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + G_ROW(r, lane);
      if (row < M) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const size_t o = (size_t)row * ldc + cb + ct * 32 + c0;
          float y = apply_act<POST_ACT>((acc[ct][r] - mean[r]) * m2[r] * gv[ct] + be[ct]);
          if (RESID) y += rs[ct][r];
          C[o] = y;
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // one row at a time: bounds the live temporaries
    }
  }
  template <class MB>
  __device__ __forceinline__ void cs(f32x16 (&acc)[4], int row0, int by, int half, int lane, const MB& mb) const {
    if (pre_out != nullptr) cs_body<true>(acc, row0, by, half, lane, mb);
    else cs_body<false>(acc, row0, by, half, lane, mb);
  }
};
