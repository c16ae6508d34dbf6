// fp32 MFMA GEMM for the backward pass:  C[M,N] (+)= alpha * A[M,K] * B[N,K]^T  over a
// K range, where either operand may be stored with the reduction index as the SLOW index
// ("k-major"):
//   dX = dY * W        A = dY   row-major (k contiguous)      B = W  k-major  (W[k][n])
//   dW = dY^T * X      A = dY   k-major   (dY[k][m])          B = X  k-major  (X[k][n])
// Same tile as the forward kernel (128 rows x 32*NT columns per workgroup, one 32-row slab
// per wave, v_mfma_f32_32x32x2_f32), plus blockIdx.z splitting the reduction: each split
// adds its partial tile with float atomics (dW sums over tens of thousands of rows).
#pragma once
#include "gemm.h"

// Operand stored [rows][K], k contiguous.
struct OpRow {
  static constexpr bool kmajor = false;
  const float* p; int ld; int rows; int K;
  __device__ __forceinline__ f32x4 load(int r, int k) const {  // 4 consecutive k
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < rows) {
      const float* q = p + (size_t)r * ld + k;
      if (((ld & 3) == 0) && k + 3 < K) {
        v = *reinterpret_cast<const f32x4*>(q);
      } else {
        if (k + 0 < K) v[0] = q[0];
        if (k + 1 < K) v[1] = q[1];
        if (k + 2 < K) v[2] = q[2];
        if (k + 3 < K) v[3] = q[3];
      }
    }
    return v;
  }
};
// Operand stored [K][cols], the tile's row index contiguous: element (c, k) at p[k*ld + c].
struct OpK {
  static constexpr bool kmajor = true;
  const float* p; int ld; int rows; int K;  // rows = number of valid c
  __device__ __forceinline__ f32x4 load(int k, int c) const {  // 4 consecutive c
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (k < K) {
      const float* q = p + (size_t)k * ld + c;
      if (((ld & 3) == 0) && c + 3 < rows) {
        v = *reinterpret_cast<const f32x4*>(q);
      } else {
        if (c + 0 < rows) v[0] = q[0];
        if (c + 1 < rows) v[1] = q[1];
        if (c + 2 < rows) v[2] = q[2];
        if (c + 3 < rows) v[3] = q[3];
      }
    }
    return v;
  }
};

#define GB_LDK_A 132   // k-major A tile row: 128 + 4 floats
template <int NT> struct GbLdkB { static constexpr int v = NT * 32 + 4; };

template <int NT, class OPA, class OPB, class EPI>
__global__ __launch_bounds__(256, 2) void gemm_bwd_kernel(OPA opa, OPB opb, int k_split, EPI epi) {
  constexpr int A_FLOATS = OPA::kmajor ? 32 * GB_LDK_A : G_BM * G_LDS;
  constexpr int B_FLOATS = OPB::kmajor ? 32 * GbLdkB<NT>::v : NT * 32 * G_LDS;
  __shared__ __attribute__((aligned(16))) float smem[A_FLOATS + B_FLOATS];
  float* sA = smem;
  float* sB = smem + A_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row_blk = blockIdx.x * G_BM;
  const int col_blk = blockIdx.y * NT * 32;
  const int k_begin = blockIdx.z * k_split;
  const int k_end = min(opa.K, k_begin + k_split);

  f32x16 acc[NT];
#pragma unroll
  for (int ct = 0; ct < NT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;

  constexpr int NB4 = OPB::kmajor ? NT : NT;  // float4 per thread for the B tile
  f32x4 ra[4], rb[NB4];
  OPA la = opa; la.K = k_end;  // loaders clip at the end of this split
  OPB lb = opb; lb.K = k_end;

  auto gload = [&](int k0) {
    if (OPA::kmajor) {
#pragma unroll
      for (int i = 0; i < 4; ++i) ra[i] = la.load(k0 + (tid >> 5) + 8 * i, row_blk + (tid & 31) * 4);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) ra[i] = la.load(row_blk + (tid >> 3) + 32 * i, k0 + (tid & 7) * 4);
    }
    if (OPB::kmajor) {
      constexpr int CPR = NT * 8;           // float4 per k-row of the B tile
      constexpr int RPP = 256 / CPR;        // k-rows per pass
#pragma unroll
      for (int i = 0; i < NT; ++i) rb[i] = lb.load(k0 + tid / CPR + RPP * i, col_blk + (tid % CPR) * 4);
    } else {
#pragma unroll
      for (int i = 0; i < NT; ++i) rb[i] = lb.load(col_blk + (tid >> 3) + 32 * i, k0 + (tid & 7) * 4);
    }
  };
  auto lds_store = [&]() {
    if (OPA::kmajor) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<f32x4*>(&sA[((tid >> 5) + 8 * i) * GB_LDK_A + (tid & 31) * 4]) = ra[i];
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<f32x4*>(&sA[((tid >> 3) + 32 * i) * G_LDS + (tid & 7) * 4]) = ra[i];
    }
    if (OPB::kmajor) {
      constexpr int CPR = NT * 8, RPP = 256 / CPR;
#pragma unroll
      for (int i = 0; i < NT; ++i)
        *reinterpret_cast<f32x4*>(&sB[(tid / CPR + RPP * i) * GbLdkB<NT>::v + (tid % CPR) * 4]) = rb[i];
    } else {
#pragma unroll
      for (int i = 0; i < NT; ++i)
        *reinterpret_cast<f32x4*>(&sB[((tid >> 3) + 32 * i) * G_LDS + (tid & 7) * 4]) = rb[i];
    }
  };

  const int fi = lane & 31, hk = lane >> 5;
  if (k_begin < k_end) gload(k_begin);
  for (int k0 = k_begin; k0 < k_end; k0 += G_BK) {
    lds_store();
    __syncthreads();
    if (k0 + G_BK < k_end) gload(k0 + G_BK);
    float a[16];
    if (OPA::kmajor) {
#pragma unroll
      for (int q = 0; q < 16; ++q) a[q] = sA[((q >> 2) * 8 + hk * 4 + (q & 3)) * GB_LDK_A + wave * 32 + fi];
    } else {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(&sA[(wave * 32 + fi) * G_LDS + kk * 8 + hk * 4]);
        a[kk * 4 + 0] = t[0]; a[kk * 4 + 1] = t[1]; a[kk * 4 + 2] = t[2]; a[kk * 4 + 3] = t[3];
      }
    }
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
      float b[16];
      if (OPB::kmajor) {
#pragma unroll
        for (int q = 0; q < 16; ++q) b[q] = sB[((q >> 2) * 8 + hk * 4 + (q & 3)) * GbLdkB<NT>::v + ct * 32 + fi];
      } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(&sB[(ct * 32 + fi) * G_LDS + kk * 8 + hk * 4]);
          b[kk * 4 + 0] = t[0]; b[kk * 4 + 1] = t[1]; b[kk * 4 + 2] = t[2]; b[kk * 4 + 3] = t[3];
        }
      }
#pragma unroll
      for (int q = 0; q < 16; ++q)
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[q], acc[ct], 0, 0, 0);
    }
    __syncthreads();
  }
  epi(acc, row_blk + wave * 32, col_blk, lane);
}

// ------------------------------------------------------------------------ epilogues
enum { MUL_NONE = 0, MUL_SILU_GRAD = 1, MUL_GELU_GRAD = 2 };

// C[row, col] (+)= alpha * acc [* act'(pre[row, col])]
template <int NT, int MUL, bool ACCUM>
struct EpiBwdStore {
  int M; int N;           // valid rows / columns of C
  float* C; int ldc;
  const float* pre; int ldp;
  float alpha;
  DropSite drop;   // thr != 0 (column-split form only): C's activation was dropout-masked in the forward
                   // pass after its nonlinearity; the gradient takes the same mask (row width N)
  __device__ __forceinline__ void operator()(f32x16 (&acc)[NT], int row0, int col_blk, int lane) const {
    const int c0 = lane & 31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + G_ROW(r, lane);
      if (row < M) {
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
          const int col = col_blk + ct * 32 + c0;
          if (col < N) {
            float v = acc[ct][r] * alpha;
            if (MUL == MUL_SILU_GRAD) v *= silu_grad(pre[(size_t)row * ldp + col]);
            if (MUL == MUL_GELU_GRAD) v *= gelu_exact_grad(pre[(size_t)row * ldp + col]);
            float* o = C + (size_t)row * ldc + col;
            if (ACCUM) *o += v; else *o = v;
          }
        }
      }
    }
  }
  // Column-split form (gemm3.h; N a multiple of 256): the values to accumulate into and the saved
  // pre-activations are loaded in one batch before the first store.
  template <class MB>
  __device__ __forceinline__ void cs(f32x16 (&acc)[4], int row0, int by, int half, int lane, const MB&) const {
    const int cb = by * 256 + half * 128 + (lane & 31);
    float old[ACCUM ? 4 : 1][16], pv[MUL != MUL_NONE ? 4 : 1][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + G_ROW(r, lane);
      const int rr = row < M ? row : M - 1;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        if (ACCUM) old[ct][r] = C[(size_t)rr * ldc + cb + ct * 32];
        if (MUL != MUL_NONE) pv[ct][r] = pre[(size_t)rr * ldp + cb + ct * 32];
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + G_ROW(r, lane);
      if (row < M) {
        f32x4 ms = {1.f, 1.f, 1.f, 1.f};
        if (drop.thr != 0u) ms = drop_scale4(drop, drop_row_quad(drop, row, cb, N));
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          float v = acc[ct][r] * alpha * ms[ct];
          if (MUL == MUL_SILU_GRAD) v *= silu_grad(pv[ct][r]);
          if (MUL == MUL_GELU_GRAD) v *= gelu_exact_grad(pv[ct][r]);
          if (ACCUM) v += old[ct][r];
          C[(size_t)row * ldc + cb + ct * 32] = v;
        }
      }
    }
  }
};

// C[row, col] += acc with float atomics (split reductions; C zeroed by the caller).
template <int NT>
struct EpiBwdAtomic {
  int M; int N;
  float* C; int ldc;
  __device__ __forceinline__ void operator()(f32x16 (&acc)[NT], int row0, int col_blk, int lane) const {
    const int c0 = lane & 31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + G_ROW(r, lane);
      if (row < M) {
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
          const int col = col_blk + ct * 32 + c0;
          if (col < N) atomicAdd(C + (size_t)row * ldc + col, acc[ct][r]);
        }
      }
    }
  }
};
