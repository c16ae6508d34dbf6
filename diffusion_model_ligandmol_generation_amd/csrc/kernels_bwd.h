// Non-GEMM kernels of the score-network backward pass.  Same mapping rules as the forward
// kernels: one wave per row / destination atom / molecule / (molecule, head); per-molecule
// reductions are done by the wave that owns the molecule (its atoms are contiguous), so
// they need no atomics and have a fixed order.  Parameter-gradient sums over all rows use a
// per-workgroup partial followed by float atomics into the (pre-zeroed) gradient tensor.
#pragma once
#include "common.h"

// out[row, off + k] = sum_o dy[row, o] * W[o, k]   (dX of a NOUT-wide final linear)
template <int NOUT>
__global__ __launch_bounds__(256) void k_small_dx(const float* __restrict__ dy, const float* __restrict__ W,
                                                  float* __restrict__ out, int ldo, int off, int M) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int o = 0; o < NOUT; ++o)
    a += dy[(size_t)row * NOUT + o] * *reinterpret_cast<const f32x4*>(W + (size_t)o * 256 + lane * 4);
  *reinterpret_cast<f32x4*>(out + (size_t)row * ldo + off + lane * 4) = a;
}

// dW[o, k] += sum_rows dy[row, o] * y[row, off + k];  db[o] += sum_rows dy[row, o]
template <int NOUT>
__global__ __launch_bounds__(256) void k_small_dw(const float* __restrict__ dy, const float* __restrict__ y, int ldy,
                                                  int off, float* __restrict__ dW, float* __restrict__ db, int M) {
  const int k = threadIdx.x;
  const int r0 = blockIdx.x * 128, r1 = min(M, r0 + 128);
  float acc[NOUT];
#pragma unroll
  for (int o = 0; o < NOUT; ++o) acc[o] = 0.f;
  float bs = 0.f;
  for (int r = r0; r < r1; ++r) {
    const float v = y[(size_t)r * ldy + off + k];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) acc[o] += dy[(size_t)r * NOUT + o] * v;
    if (k < NOUT) bs += dy[(size_t)r * NOUT + k];
  }
#pragma unroll
  for (int o = 0; o < NOUT; ++o) atomicAdd(dW + (size_t)o * 256 + k, acc[o]);
  if (k < NOUT) atomicAdd(db + k, bs);
}

// out[c] += sum_rows Y[row, c] * (rowscale ? rowscale[row] : 1)
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ Y, int ld, int M, int ncols,
                                                const float* __restrict__ rowscale, float* __restrict__ out) {
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= ncols) return;
  const int r0 = blockIdx.x * 128, r1 = min(M, r0 + 128);
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += Y[(size_t)r * ld + c] * (rowscale != nullptr ? rowscale[r] : 1.0f);
  atomicAdd(out + c, s);
}

// Weight gradient for shapes the LDS-DMA kernel cannot take (widths 32 / 119, unaligned leading
// dimensions) with a long reduction:  dW[n, k] += sum_e dY[e, n] X[e, k]  (+ db[n] += sum_e dY[e, n]).
// One wave per (32-row n tile, 32-column k tile, slice of `rows_per_slice` reduction rows), fp32 MFMA with
// both operands loaded straight from global memory: for step (e, e + 1) lane (i, h) holds
// dY[e + h, n0 + i] and X[e + h, k0 + i] - two coalesced 128-B row segments per half-wave, no alignment
// needed.  Columns beyond nout / kin read as zero.  4 waves per workgroup.
__global__ __launch_bounds__(256) void k_dw_direct(const float* __restrict__ dY, int lddy, int nout,
                                                   const float* __restrict__ X, int ldx, int kin, int rows,
                                                   int rows_per_slice, float* __restrict__ dW, int lddw,
                                                   float* __restrict__ db) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_tiles = (nout + 31) >> 5, k_tiles = (kin + 31) >> 5;
  const int item = blockIdx.x * 4 + wave;
  const int tile = item % (n_tiles * k_tiles), slice = item / (n_tiles * k_tiles);
  const int nt = tile / k_tiles, kt = tile % k_tiles;
  const int e0 = slice * rows_per_slice, e1 = min(rows, e0 + rows_per_slice);
  if (e0 >= e1) return;
  const int i = lane & 31, h = lane >> 5;
  const bool n_ok = nt * 32 + i < nout, k_ok = kt * 32 + i < kin;
  const float* yp = dY + (n_ok ? nt * 32 + i : 0);
  const float* xp = X + (k_ok ? kt * 32 + i : 0);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bsum = 0.f;
  int e = e0;
  for (; e + 32 <= e1; e += 32) {   // sixteen MFMA steps per pass, their 32 loads issued together (latency-bound otherwise)
    float a[16], b[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      a[u] = yp[(size_t)(e + 2 * u + h) * lddy];
      b[u] = xp[(size_t)(e + 2 * u + h) * ldx];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      a[u] = n_ok ? a[u] : 0.f;
      b[u] = k_ok ? b[u] : 0.f;
      bsum += a[u];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
    }
  }
  for (; e < e1; e += 2) {
    const bool ok = e + h < e1;
    const float a = (ok && n_ok) ? yp[(size_t)(e + h) * lddy] : 0.f;
    const float b = (ok && k_ok) ? xp[(size_t)(e + h) * ldx] : 0.f;
    bsum += a;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  // accumulator: column = k (lane & 31), row = n offset (r & 3) + 8 (r >> 2) + 4 h
  if (k_ok) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (n < nout) atomicAdd(dW + (size_t)n * lddw + kt * 32 + i, acc[r]);
    }
  }
  if (db != nullptr && kt == 0) {
    bsum += __shfl_xor(bsum, 32);
    if (h == 0 && n_ok) atomicAdd(db + nt * 32 + i, bsum);
  }
}

// The three 32-wide column sums of the radial MLP backward in one pass over the edges (model1.py:56-60):
//   out_b2[c] += sum_e drbf[e, c];  out_w0[c] += sum_e dr1pre[e, c] dist[e];  out_b0[c] += sum_e dr1pre[e, c]
// 256 threads = 8 row groups x 32 columns over 256 edges per workgroup.
__global__ __launch_bounds__(256) void k_radial_colsums(const float* __restrict__ drbf, const float* __restrict__ dr1pre,
                                                        const float* __restrict__ dist, float* __restrict__ out_b2,
                                                        float* __restrict__ out_w0, float* __restrict__ out_b0, int E) {
  __shared__ float sh[3][8][32];
  const int c = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int r0 = blockIdx.x * 256, r1 = min(E, r0 + 256);
  float s2 = 0.f, sw = 0.f, sb = 0.f;
  for (int r = r0 + grp; r < r1; r += 8) {
    const float d1 = dr1pre[(size_t)r * 32 + c];
    s2 += drbf[(size_t)r * 32 + c];
    sw += d1 * dist[r];
    sb += d1;
  }
  sh[0][grp][c] = s2; sh[1][grp][c] = sw; sh[2][grp][c] = sb;
  __syncthreads();
  if (threadIdx.x < 96) {
    const int which = threadIdx.x >> 5;
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) t += sh[which][g][c];
    atomicAdd((which == 0 ? out_b2 : which == 1 ? out_w0 : out_b0) + c, t);
  }
}

// LayerNorm backward over 256-wide rows, one wave per row, `rpw` rows per wave (each workgroup ends with
// 512 atomics into the same dgamma / dbeta addresses, which serialise in L2: large M takes more rows
// per wave so that the atomic chain stays shorter than the streaming time).
//   v = pre(u), xh = (v - mean) rstd, y = post(xh g + b)
//   dy_total = dy[row] (+ dy2[idx[row]])
// writes du[row] and accumulates dgamma / dbeta.
template <int PRE_ACT, int POST_ACT>
__global__ __launch_bounds__(256) void k_ln_bwd(const float* __restrict__ u, int ldu, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, float eps,
                                                const float* __restrict__ dy, int ldy,
                                                const float* __restrict__ dy2, int ld2, const int32_t* __restrict__ idx,
                                                float* __restrict__ du, int ldo, float* __restrict__ dgamma,
                                                float* __restrict__ dbeta, int M, int rpw) {
  __shared__ float sh[2][4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + lane * 4);
  const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + lane * 4);
  f32x4 dg = {0.f, 0.f, 0.f, 0.f}, db = {0.f, 0.f, 0.f, 0.f};
  const int rbase = (blockIdx.x * 4 + wave) * rpw;
  // rows in groups of four, the group's loads issued before any of its arithmetic (a row is otherwise a
  // serial chain of one memory latency and five wave reductions)
  for (int rr = 0; rr < rpw; rr += 4) {
    f32x4 uu4[4], dy4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = rbase + rr + q;
      const int rc = row < M ? row : M - 1;
      uu4[q] = *reinterpret_cast<const f32x4*>(u + (size_t)rc * ldu + lane * 4);
      dy4[q] = *reinterpret_cast<const f32x4*>(dy + (size_t)rc * ldy + lane * 4);
      if (dy2 != nullptr) dy4[q] += *reinterpret_cast<const f32x4*>(dy2 + (size_t)idx[rc] * ld2 + lane * 4);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = rbase + rr + q;
      if (row >= M || rr + q >= rpw) break;
      const f32x4 uu = uu4[q];
      f32x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = PRE_ACT == 1 ? gelu_exact(uu[i]) : uu[i];
      const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / 256.0f);
      const f32x4 d = v - mean;
      const float rstd = 1.0f / sqrtf(wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.0f / 256.0f) + eps);
      const f32x4 xh = d * rstd;
      f32x4 dyv = dy4[q];
      if (POST_ACT == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dyv[i] *= gelu_exact_grad(xh[i] * g[i] + bt[i]);
      }
      dg += dyv * xh;
      db += dyv;
      const f32x4 dxh = dyv * g;
      const float m1 = wave_sum((dxh[0] + dxh[1]) + (dxh[2] + dxh[3])) * (1.0f / 256.0f);
      const float m2 = wave_sum((dxh[0] * xh[0] + dxh[1] * xh[1]) + (dxh[2] * xh[2] + dxh[3] * xh[3])) * (1.0f / 256.0f);
      f32x4 dv = (dxh - m1 - xh * m2) * rstd;
      if (PRE_ACT == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dv[i] *= gelu_exact_grad(uu[i]);
      }
      *reinterpret_cast<f32x4*>(du + (size_t)row * ldo + lane * 4) = dv;
    }
  }
  *reinterpret_cast<f32x4*>(&sh[0][wave][lane * 4]) = dg;
  *reinterpret_cast<f32x4*>(&sh[1][wave][lane * 4]) = db;
  __syncthreads();
  const int t = threadIdx.x;
  atomicAdd(dgamma + t, (sh[0][0][t] + sh[0][1][t]) + (sh[0][2][t] + sh[0][3][t]));
  atomicAdd(dbeta + t, (sh[1][0][t] + sh[1][1][t]) + (sh[1][2][t] + sh[1][3][t]));
}

// Gated residual backward (model1.py:295 / :300), one WORKGROUP per molecule (its atoms strided over the
// four waves, per-molecule sums combined through LDS in a fixed order: no atomics, bit-reproducible):
//   dproj[n] = dx[n] * gate[mol];   dgate[mol] = sum_n dx[n] * proj[n]
__global__ __launch_bounds__(256) void k_gate_bwd(const float* __restrict__ dx, const float* __restrict__ proj,
                                                  const float* __restrict__ mod, int mod_stride, int gate_off,
                                                  const int32_t* __restrict__ mol_ptr, int B, float* __restrict__ dproj,
                                                  float* __restrict__ dmod, DropSite drop = DropSite{}) {
  // drop.thr != 0: proj is the dropout-masked projection the forward pass saved; dproj takes the mask
  __shared__ float sh[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mol = blockIdx.x;
  const f32x4 gate = *reinterpret_cast<const f32x4*>(mod + (size_t)mol * mod_stride + gate_off + lane * 4);
  f32x4 dgate = {0.f, 0.f, 0.f, 0.f};
  for (int a = mol_ptr[mol] + wave; a < mol_ptr[mol + 1]; a += 4) {
    const f32x4 d = *reinterpret_cast<const f32x4*>(dx + (size_t)a * 256 + lane * 4);
    dgate += d * *reinterpret_cast<const f32x4*>(proj + (size_t)a * 256 + lane * 4);
    f32x4 dp = d * gate;
    if (drop.thr != 0u) {
#pragma unroll
      for (int e = 0; e < 4; ++e) dp[e] *= drop_row_scale(drop, a, lane * 4 + e, 256);
    }
    *reinterpret_cast<f32x4*>(dproj + (size_t)a * 256 + lane * 4) = dp;
  }
  *reinterpret_cast<f32x4*>(&sh[wave][lane * 4]) = dgate;
  __syncthreads();
  const int t = threadIdx.x;
  dmod[(size_t)mol * 1536 + gate_off + t] = (sh[0][t] + sh[1][t]) + (sh[2][t] + sh[3][t]);
}

// Modulated pre-norm backward (model1.py:293 / :298), one workgroup per molecule (as k_gate_bwd):
//   xn = LN(x)(1 + scale) + shift  ->  dx_out[n] = dres[n] + LNbwd(dxn[n] (1 + scale)),
//   dshift[mol] = sum_n dxn[n],  dscale[mol] = sum_n dxn[n] * xhat[n]
__global__ __launch_bounds__(256) void k_lnmod_bwd(const float* __restrict__ x, const float* __restrict__ dxn,
                                                   const float* __restrict__ dres, const float* __restrict__ mod,
                                                   int mod_stride, int shift_off, int scale_off,
                                                   const int32_t* __restrict__ mol_ptr, int B,
                                                   float* __restrict__ dx_out, float* __restrict__ dmod) {
  __shared__ float sh[2][4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mol = blockIdx.x;
  const f32x4 sc1 = 1.0f + *reinterpret_cast<const f32x4*>(mod + (size_t)mol * mod_stride + scale_off + lane * 4);
  f32x4 dsh = {0.f, 0.f, 0.f, 0.f}, dsc = {0.f, 0.f, 0.f, 0.f};
  for (int a = mol_ptr[mol] + wave; a < mol_ptr[mol + 1]; a += 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)a * 256 + lane * 4);
    const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / 256.0f);
    const f32x4 d = v - mean;
    const float rstd = 1.0f / sqrtf(wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.0f / 256.0f) + 1e-6f);
    const f32x4 xh = d * rstd;
    const f32x4 dy = *reinterpret_cast<const f32x4*>(dxn + (size_t)a * 256 + lane * 4);
    dsh += dy;
    dsc += dy * xh;
    const f32x4 dxh = dy * sc1;
    const float m1 = wave_sum((dxh[0] + dxh[1]) + (dxh[2] + dxh[3])) * (1.0f / 256.0f);
    const float m2 = wave_sum((dxh[0] * xh[0] + dxh[1] * xh[1]) + (dxh[2] * xh[2] + dxh[3] * xh[3])) * (1.0f / 256.0f);
    f32x4 o = (dxh - m1 - xh * m2) * rstd;
    if (dres != nullptr) o += *reinterpret_cast<const f32x4*>(dres + (size_t)a * 256 + lane * 4);
    *reinterpret_cast<f32x4*>(dx_out + (size_t)a * 256 + lane * 4) = o;
  }
  *reinterpret_cast<f32x4*>(&sh[0][wave][lane * 4]) = dsh;
  *reinterpret_cast<f32x4*>(&sh[1][wave][lane * 4]) = dsc;
  __syncthreads();
  const int t = threadIdx.x;
  dmod[(size_t)mol * 1536 + shift_off + t] = (sh[0][0][t] + sh[0][1][t]) + (sh[0][2][t] + sh[0][3][t]);
  dmod[(size_t)mol * 1536 + scale_off + t] = (sh[1][0][t] + sh[1][1][t]) + (sh[1][2][t] + sh[1][3][t]);
}

// Attention backward for one (molecule, head) per wave (blockDim.x / 64 heads per workgroup).
// Recomputes P from q, k and the saved log-sum-exp.  Pass 1 (lane = query i): delta_i and
// dq_i.  Pass 2 (lane = key j): dk_j and dv_j.  LDS per wave: q, k, v, dO rows (4 x n x 32)
// plus lse and delta (2 x n).
__global__ __launch_bounds__(256) void k_attention_bwd(const float* __restrict__ qkv, const float* __restrict__ o,
                                                       const float* __restrict__ d_o, const float* __restrict__ lse,
                                                       const int32_t* __restrict__ mol_ptr, float* __restrict__ dqkv,
                                                       int max_n) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hpb = blockDim.x >> 6, bpm = 8 / hpb;   // heads per workgroup, workgroups per molecule
  const int mol = blockIdx.x / bpm, head = (blockIdx.x % bpm) * hpb + wave;
  const int a0 = mol_ptr[mol], n = mol_ptr[mol + 1] - a0;
  float* base = smem + (size_t)wave * (max_n * 130);
  float* sq = base;                 // [n][32]
  float* sk = sq + max_n * 32;
  float* sv = sk + max_n * 32;
  float* sdo = sv + max_n * 32;
  float* slse = sdo + max_n * 32;   // [n]
  float* sdel = slse + max_n;       // [n]
  for (int i = lane; i < n * 8; i += 64) {
    const int j = i >> 3, c = (i & 7) * 4;
    const float* r = qkv + (size_t)(a0 + j) * 768 + head * 32 + c;
    *reinterpret_cast<f32x4*>(sq + j * 32 + c) = *reinterpret_cast<const f32x4*>(r);
    *reinterpret_cast<f32x4*>(sk + j * 32 + c) = *reinterpret_cast<const f32x4*>(r + 256);
    *reinterpret_cast<f32x4*>(sv + j * 32 + c) = *reinterpret_cast<const f32x4*>(r + 512);
    *reinterpret_cast<f32x4*>(sdo + j * 32 + c) =
        *reinterpret_cast<const f32x4*>(d_o + (size_t)(a0 + j) * 256 + head * 32 + c);
  }
  for (int i = lane; i < n; i += 64) {
    slse[i] = lse[(size_t)(a0 + i) * 8 + head];
    float dl = 0.f;
    for (int c = 0; c < 32; ++c)
      dl += d_o[(size_t)(a0 + i) * 256 + head * 32 + c] * o[(size_t)(a0 + i) * 256 + head * 32 + c];
    sdel[i] = dl;
  }
  __syncthreads();
  const float scale = 0.17677669529663687f;
  for (int i = lane; i < n; i += 64) {   // dq_i = scale * sum_j ds_ij k_j
    float q[32], dov[32], dq[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) { q[c] = sq[i * 32 + c]; dov[c] = sdo[i * 32 + c]; dq[c] = 0.f; }
    const float li = slse[i], di = sdel[i];
    for (int j = 0; j < n; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int c = 0; c < 32; ++c) { s += q[c] * sk[j * 32 + c]; dp += dov[c] * sv[j * 32 + c]; }
      const float p = expf(s * scale - li);
      const float ds = p * (dp - di) * scale;
#pragma unroll
      for (int c = 0; c < 32; ++c) dq[c] += ds * sk[j * 32 + c];
    }
    float* out = dqkv + (size_t)(a0 + i) * 768 + head * 32;
#pragma unroll
    for (int c = 0; c < 32; ++c) out[c] = dq[c];
  }
  for (int j = lane; j < n; j += 64) {   // dk_j = scale * sum_i ds_ij q_i ; dv_j = sum_i p_ij dO_i
    float k[32], v[32], dk[32], dv[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) { k[c] = sk[j * 32 + c]; v[c] = sv[j * 32 + c]; dk[c] = 0.f; dv[c] = 0.f; }
    for (int i = 0; i < n; ++i) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int c = 0; c < 32; ++c) { s += sq[i * 32 + c] * k[c]; dp += sdo[i * 32 + c] * v[c]; }
      const float p = expf(s * scale - slse[i]);
      const float ds = p * (dp - sdel[i]) * scale;
#pragma unroll
      for (int c = 0; c < 32; ++c) { dk[c] += ds * sq[i * 32 + c]; dv[c] += p * sdo[i * 32 + c]; }
    }
    float* out = dqkv + (size_t)(a0 + j) * 768 + head * 32;
#pragma unroll
    for (int c = 0; c < 32; ++c) { out[256 + c] = dk[c]; out[512 + c] = dv[c]; }
  }
}

// MFMA form of the attention backward (fp32 v_mfma_f32_32x32x2_f32), one wave per (molecule, head),
// WPB waves per workgroup, TK = number of 32-atom tiles (n <= 32 TK).  q, k, v, dO of the head sit in
// LDS as [n][33] (zero rows beyond n), lse and delta_i = dO_i . O_i as [n].  As in the forward
// kernel a product whose result feeds the next product is computed TRANSPOSED, so that its
// accumulator (lane = column, register = row) can be used directly as the B operand:
//   pass 1, lane = query i:  S^T = K Q'^T, dP^T = V dO^T  ->  dS^T = P^T (dP^T - delta_i)
//                            dQ^T[d, i] = sum_j K[j, d] dS^T[j, i]
//   pass 2, lane = key j:    S = Q' K^T, dP = dO V^T      ->  P, dS (row statistics read per register)
//                            dV^T[d, j] = sum_i dO[i, d] P[i, j],  dK^T[d, j] = sum_i Q[i, d] dS[i, j]
// with Q' = Q / sqrt(32); dQ and dK carry the same factor.  Register r of lane half h is row
// (r & 3) + 8 (r >> 2) + 4 h of a tile.
#define GB_ROW(r, h) (((r) & 3) + 8 * ((r) >> 2) + 4 * (h))
// DROP: the forward pass multiplied the probabilities by the dropout mask Ms of site `drop`
// (O = (P Ms) V): dV takes P Ms, dP = (dO V^T) Ms, and delta_i = dO_i . O_i is unchanged in form.
// PAIR: two waves per (molecule, head) - they fill the LDS images together, then one runs pass 1 and the other
// pass 2 (the passes only read the images).  The images of a 64-atom head are 34 KiB, so a CU holds four of them:
// with one wave per image each SIMD has a single wave whose loads, LDS reads, exponentials and MFMAs run one
// after the other; with two, one wave's MFMAs cover the other's waits.  Every output element is computed by the
// same instructions in the same order either way.
template <int TK, int WPB, bool DROP = false, bool PAIR = true>
__global__ __launch_bounds__(WPB * 64 * (PAIR ? 2 : 1)) void k_attention_bwd_mfma(const float* __restrict__ qkv, const float* __restrict__ o,
                                                                 const float* __restrict__ d_o, const float* __restrict__ lse,
                                                                 const int32_t* __restrict__ mol_ptr,
                                                                 float* __restrict__ dqkv, int n_mols,
                                                                 DropSite drop = DropSite{}) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NP = TK * 32;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wave = PAIR ? wv % WPB : wv;     // image slot of the workgroup
  const int role = PAIR ? wv / WPB : 0;      // PAIR: 0 runs pass 1, 1 runs pass 2
  const int item = blockIdx.x * WPB + wave;
  const int mol = item >> 3, head = item & 7;
  float* sq = smem + (size_t)wave * (NP * 134);
  float* sk = sq + NP * 33;
  float* sv = sk + NP * 33;
  float* sdo = sv + NP * 33;
  float* slse = sdo + NP * 33;
  float* sdel = slse + NP;
  int a0 = 0, n = 0;
  if (mol < n_mols) { a0 = mol_ptr[mol]; n = mol_ptr[mol + 1] - a0; }
  for (int i = lane + 64 * role; i < NP * 8; i += PAIR ? 128 : 64) {   // 8 lanes per row, 4 columns each
    const int j = i >> 3, c = (i & 7) * 4;
    f32x4 q4 = {0.f, 0.f, 0.f, 0.f}, k4 = q4, v4 = q4, d4 = q4, o4 = q4;
    if (j < n) {
      const float* r = qkv + (size_t)(a0 + j) * 768 + head * 32 + c;
      q4 = *reinterpret_cast<const f32x4*>(r);
      k4 = *reinterpret_cast<const f32x4*>(r + 256);
      v4 = *reinterpret_cast<const f32x4*>(r + 512);
      d4 = *reinterpret_cast<const f32x4*>(d_o + (size_t)(a0 + j) * 256 + head * 32 + c);
      o4 = *reinterpret_cast<const f32x4*>(o + (size_t)(a0 + j) * 256 + head * 32 + c);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sq[j * 33 + c + e] = q4[e]; sk[j * 33 + c + e] = k4[e]; sv[j * 33 + c + e] = v4[e]; sdo[j * 33 + c + e] = d4[e];
    }
    float dl = (d4[0] * o4[0] + d4[1] * o4[1]) + (d4[2] * o4[2] + d4[3] * o4[3]);
    dl += __shfl_xor(dl, 1); dl += __shfl_xor(dl, 2); dl += __shfl_xor(dl, 4);
    if ((i & 7) == 0) {
      sdel[j] = dl;
      slse[j] = j < n ? lse[(size_t)(a0 + j) * 8 + head] : 0.f;
    }
  }
  __syncthreads();
  if (mol >= n_mols) return;
  const int fi = lane & 31, h = lane >> 5;
  const float scale = 0.17677669529663687f;
  // ---------------- pass 1: lane = query
#pragma unroll 1
  for (int it = 0; it < TK; ++it) {
    if (it * 32 >= n || (PAIR && role != 0)) break;
    const int qi = it * 32 + fi;
    float qreg[16], doreg[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) { qreg[s] = sq[qi * 33 + 2 * s + h] * scale; doreg[s] = sdo[qi * 33 + 2 * s + h]; }
    const float lse_i = slse[qi], del_i = sdel[qi];
    f32x16 dq;
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[r] = 0.f;
#pragma unroll 1
    for (int jt = 0; jt < TK; ++jt) {
      if (jt * 32 >= n) break;
      f32x16 st, dpt;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = 0.f; dpt[r] = 0.f; }
      const float* krow = sk + (jt * 32 + fi) * 33 + h;
      const float* vrow = sv + (jt * 32 + fi) * 33 + h;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        st = __builtin_amdgcn_mfma_f32_32x32x2f32(krow[2 * s], qreg[s], st, 0, 0, 0);
        dpt = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[2 * s], doreg[s], dpt, 0, 0, 0);
      }
      if (DROP) {   // registers 4g..4g+3 are four consecutive keys: one mask quad
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const f32x4 ms = drop_scale4(drop, drop_att_quad(drop, a0 + qi, head, jt * 32 + gq * 8 + h * 4));
#pragma unroll
          for (int e = 0; e < 4; ++e) dpt[gq * 4 + e] *= ms[e];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = jt * 32 + GB_ROW(r, h);
        const float p = (j < n && qi < n) ? __expf(st[r] - lse_i) : 0.f;
        st[r] = p * (dpt[r] - del_i);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r)
        dq = __builtin_amdgcn_mfma_f32_32x32x2f32(sk[(jt * 32 + GB_ROW(r, h)) * 33 + fi], st[r], dq, 0, 0, 0);
    }
    if (qi < n) {
      float* out = dqkv + (size_t)(a0 + qi) * 768 + head * 32;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {   // registers 4g..4g+3 are d = 8g + 4h + (0..3)
        const f32x4 v = {dq[gq * 4 + 0] * scale, dq[gq * 4 + 1] * scale, dq[gq * 4 + 2] * scale, dq[gq * 4 + 3] * scale};
        *reinterpret_cast<f32x4*>(out + gq * 8 + h * 4) = v;
      }
    }
  }
  // ---------------- pass 2: lane = key
#pragma unroll 1
  for (int jt = 0; jt < TK; ++jt) {
    if (jt * 32 >= n || (PAIR && role != 1)) break;
    const int kj = jt * 32 + fi;
    float kreg[16], vreg[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) { kreg[s] = sk[kj * 33 + 2 * s + h]; vreg[s] = sv[kj * 33 + 2 * s + h]; }
    f32x16 dk, dv;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }
#pragma unroll 1
    for (int it = 0; it < TK; ++it) {
      if (it * 32 >= n) break;
      f32x16 sm, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sm[r] = 0.f; dp[r] = 0.f; }
      const float* qrow = sq + (it * 32 + fi) * 33 + h;
      const float* drow = sdo + (it * 32 + fi) * 33 + h;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        sm = __builtin_amdgcn_mfma_f32_32x32x2f32(qrow[2 * s] * scale, kreg[s], sm, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(drow[2 * s], vreg[s], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = it * 32 + GB_ROW(r, h);
        const float p = (i < n && kj < n) ? __expf(sm[r] - slse[i]) : 0.f;
        float ms = 1.0f;
        if (DROP) ms = drop_scale4(drop, drop_att_quad(drop, a0 + i, head, kj))[kj & 3];
        sm[r] = p * ms;
        dp[r] = p * (dp[r] * ms - sdel[i]);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = it * 32 + GB_ROW(r, h);
        dv = __builtin_amdgcn_mfma_f32_32x32x2f32(sdo[i * 33 + fi], sm[r], dv, 0, 0, 0);
        dk = __builtin_amdgcn_mfma_f32_32x32x2f32(sq[i * 33 + fi], dp[r], dk, 0, 0, 0);
      }
    }
    if (kj < n) {
      float* out = dqkv + (size_t)(a0 + kj) * 768 + head * 32;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const f32x4 kv = {dk[gq * 4 + 0] * scale, dk[gq * 4 + 1] * scale, dk[gq * 4 + 2] * scale, dk[gq * 4 + 3] * scale};
        const f32x4 vv = {dv[gq * 4 + 0], dv[gq * 4 + 1], dv[gq * 4 + 2], dv[gq * 4 + 3]};
        *reinterpret_cast<f32x4*>(out + 256 + gq * 8 + h * 4) = kv;
        *reinterpret_cast<f32x4*>(out + 512 + gq * 8 + h * 4) = vv;
      }
    }
  }
}

// Coordinate-gate backward (model1.py:106-112), one wave per edge, strided over the grid:
//   dwrel = 0.1 (G[dst] - G[src]);  dcw = dwrel . rel;  ds = dcw (1 - cw^2)
//   drel  = dwrel cw;  dhc_pre[j] = ds w2[j] silu'(hc_pre[j]);  dw2[j] += ds silu(hc_pre[j]);  db2 += ds
__global__ __launch_bounds__(256) void k_coord_bwd(const float* __restrict__ G, const int32_t* __restrict__ src,
                                                   const int32_t* __restrict__ dst, const float* __restrict__ rel,
                                                   const float* __restrict__ cw, const float* __restrict__ hc_pre,
                                                   const float* __restrict__ w2, float* __restrict__ drel,
                                                   float* __restrict__ dhc_pre, float* __restrict__ dw2,
                                                   float* __restrict__ db2, int E) {
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
  const float w2a = w2[lane], w2b = w2[lane + 64];
  float acc_a = 0.f, acc_b = 0.f, acc_b2 = 0.f;
  // Four edges per pass, every load of the pass issued before its first store: an edge is a chain of
  // dependent loads (indices -> G rows), and a store through drel / dhc_pre would otherwise fence the
  // next edge's loads behind it.
  for (int e0 = gw * 4; e0 < E; e0 += nw * 4) {
    float dwv[4][3], relv[4][3], cwv[4], ha[4], hb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u < E ? e0 + u : E - 1;
      const int s = src[e], d = dst[e];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        dwv[u][c] = 0.1f * (G[(size_t)d * 3 + c] - G[(size_t)s * 3 + c]);
        relv[u][c] = rel[(size_t)e * 4 + c];
      }
      cwv[u] = cw[e];
      ha[u] = hc_pre[(size_t)e * 128 + lane];
      hb[u] = hc_pre[(size_t)e * 128 + lane + 64];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u;
      if (e >= E) break;
      const float dcw = (dwv[u][0] * relv[u][0] + dwv[u][1] * relv[u][1]) + dwv[u][2] * relv[u][2];
      const float c_w = cwv[u];
      const float ds = dcw * (1.0f - c_w * c_w);
      if (lane < 3) drel[(size_t)e * 4 + lane] = dwv[u][lane == 0 ? 0 : lane == 1 ? 1 : 2] * c_w;
      if (lane == 3) drel[(size_t)e * 4 + 3] = 0.f;
      dhc_pre[(size_t)e * 128 + lane] = ds * w2a * silu_grad(ha[u]);
      dhc_pre[(size_t)e * 128 + lane + 64] = ds * w2b * silu_grad(hb[u]);
      acc_a += ds * silu(ha[u]);
      acc_b += ds * silu(hb[u]);
      acc_b2 += ds;
    }
  }
  // one set of atomics per workgroup: thousands of waves adding into the same 129 addresses serialise
  // in L2 (the kernel was bound by exactly that)
  __shared__ float red[4][129];
  const int wave = threadIdx.x >> 6;
  red[wave][lane] = acc_a;
  red[wave][lane + 64] = acc_b;
  if (lane == 0) red[wave][128] = acc_b2;
  __syncthreads();
  if (threadIdx.x < 129) {
    const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    atomicAdd(threadIdx.x < 128 ? dw2 + threadIdx.x : db2, t);
  }
}

// dP[n, 0:256] = sum over edges LEAVING n of dpre[e];  dP[n, 256:512] = sum over edges
// ENTERING n (contiguous CSR rows) - the transpose of the two gathers of the first edge linear.
__global__ __launch_bounds__(256) void k_dp_gather(const float* __restrict__ dpre, const int32_t* __restrict__ dst_ptr,
                                                   const int32_t* __restrict__ src_ptr,
                                                   const int32_t* __restrict__ src_slot, float* __restrict__ dP, int N) {
  const int lane = threadIdx.x & 63;
  const int node = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (node >= N) return;
  f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
  for (int k = src_ptr[node]; k < src_ptr[node + 1]; ++k)
    a += *reinterpret_cast<const f32x4*>(dpre + (size_t)src_slot[k] * 256 + lane * 4);
  for (int e = dst_ptr[node]; e < dst_ptr[node + 1]; ++e)
    b += *reinterpret_cast<const f32x4*>(dpre + (size_t)e * 256 + lane * 4);
  *reinterpret_cast<f32x4*>(dP + (size_t)node * 512 + lane * 4) = a;
  *reinterpret_cast<f32x4*>(dP + (size_t)node * 512 + 256 + lane * 4) = b;
}

// Radial MLP + distance backward (model1.py:88-93), one lane per edge:
//   dr1 = drbf W2;  dr1pre = dr1 silu'(r1pre);  ddist = dr1pre . w0
//   drel += ddist (rel + 1e-8) / dist
// also writes r1 = silu(r1pre) (operand of dW2) and dr1pre.
__global__ __launch_bounds__(256) void k_radial_bwd(const float* __restrict__ drbf, const float* __restrict__ r1pre,
                                                    const float* __restrict__ dist, const float* __restrict__ rel,
                                                    const float* __restrict__ w0, const float* __restrict__ W2,
                                                    float* __restrict__ drel, float* __restrict__ dr1pre,
                                                    float* __restrict__ r1, int E) {
  __shared__ float sW2[32 * 33];
  __shared__ float sw0[32];
  for (int i = threadIdx.x; i < 1024; i += 256) sW2[(i >> 5) * 33 + (i & 31)] = W2[i];
  if (threadIdx.x < 32) sw0[threadIdx.x] = w0[threadIdx.x];
  __syncthreads();
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  float dr[32];
#pragma unroll
  for (int jo = 0; jo < 32; ++jo) dr[jo] = drbf[(size_t)e * 32 + jo];
  float dd = 0.f;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    float a = 0.f;
#pragma unroll
    for (int jo = 0; jo < 32; ++jo) a += dr[jo] * sW2[jo * 33 + j];
    const float pre = r1pre[(size_t)e * 32 + j];
    const float g = a * silu_grad(pre);
    dr1pre[(size_t)e * 32 + j] = g;
    r1[(size_t)e * 32 + j] = silu(pre);
    dd += g * sw0[j];
  }
  const float inv = dd / dist[e];
#pragma unroll
  for (int c = 0; c < 3; ++c) drel[(size_t)e * 4 + c] += inv * (rel[(size_t)e * 4 + c] + 1e-8f);
}

// G[i] += sum_{e in in(i)} drel[e] - sum_{e in out(i)} drel[e]   (rel = pos[dst] - pos[src])
__global__ __launch_bounds__(256) void k_pos_scatter_bwd(const float* __restrict__ drel,
                                                         const int32_t* __restrict__ dst_ptr,
                                                         const int32_t* __restrict__ src_ptr,
                                                         const int32_t* __restrict__ src_slot, float* __restrict__ G,
                                                         int N) {
  const int lane = threadIdx.x & 63;
  const int node = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (node >= N) return;
  const int c = lane & 3;
  float d = 0.f;
  for (int k = dst_ptr[node] + (lane >> 2); k < dst_ptr[node + 1]; k += 16) d += drel[(size_t)k * 4 + c];
  for (int k = src_ptr[node] + (lane >> 2); k < src_ptr[node + 1]; k += 16) d -= drel[(size_t)src_slot[k] * 4 + c];
  d += __shfl_xor(d, 4);
  d += __shfl_xor(d, 8);
  d += __shfl_xor(d, 16);
  d += __shfl_xor(d, 32);
  if (lane < 3) G[(size_t)node * 3 + lane] += d;
}

// out[mol, :] = sum over the molecule's atoms of y[n, :]
__global__ __launch_bounds__(256) void k_mol_sum(const float* __restrict__ y, const int32_t* __restrict__ mol_ptr,
                                                 int B, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int mol = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (mol >= B) return;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int a = mol_ptr[mol]; a < mol_ptr[mol + 1]; ++a) s += *reinterpret_cast<const f32x4*>(y + (size_t)a * 256 + lane * 4);
  *reinterpret_cast<f32x4*>(out + (size_t)mol * 256 + lane * 4) = s;
}

// dx[n, :] += dpooled[mol, :] / n_mol   (backward of global_mean_pool, model1.py:745)
__global__ __launch_bounds__(256) void k_pool_bwd(const float* __restrict__ dpooled, const int32_t* __restrict__ mol_ptr,
                                                  const int32_t* __restrict__ atom_mol, float* __restrict__ dx, int N) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const int m = atom_mol[row];
  const float inv = 1.0f / (float)(mol_ptr[m + 1] - mol_ptr[m]);
  f32x4* p = reinterpret_cast<f32x4*>(dx + (size_t)row * 256 + lane * 4);
  *p += *reinterpret_cast<const f32x4*>(dpooled + (size_t)m * 256 + lane * 4) * inv;
}

// u_p[n, c] = pos[n] . Wp[c] + bp[c]   (recomputed pre-LayerNorm value of the position embedding)
__global__ __launch_bounds__(256) void k_pos_embed_pre(const float* __restrict__ pos, const float* __restrict__ Wp,
                                                       const float* __restrict__ bp, float* __restrict__ up, int N) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const float p0 = pos[(size_t)row * 3], p1 = pos[(size_t)row * 3 + 1], p2 = pos[(size_t)row * 3 + 2];
  f32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane * 4 + i;
    o[i] = (p0 * Wp[c * 3 + 0] + p1 * Wp[c * 3 + 1]) + p2 * Wp[c * 3 + 2] + bp[c];
  }
  *reinterpret_cast<f32x4*>(up + (size_t)row * 256 + lane * 4) = o;
}

// dWp[c, j] += sum_rows du[row, c] pos[row, j]   (Linear 3 -> 256)
__global__ __launch_bounds__(256) void k_pos_embed_dw(const float* __restrict__ du, const float* __restrict__ pos,
                                                      float* __restrict__ dWp, int N) {
  const int c = threadIdx.x;
  const int r0 = blockIdx.x * 128, r1 = min(N, r0 + 128);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int r = r0; r < r1; ++r) {
    const float d = du[(size_t)r * 256 + c];
    a0 += d * pos[(size_t)r * 3 + 0]; a1 += d * pos[(size_t)r * 3 + 1]; a2 += d * pos[(size_t)r * 3 + 2];
  }
  atomicAdd(dWp + c * 3 + 0, a0); atomicAdd(dWp + c * 3 + 1, a1); atomicAdd(dWp + c * 3 + 2, a2);
}

// dtemb = dtemb_atoms + dstemb * silu'(temb)   (t_emb feeds the embedding sum and, through
// SiLU, every adaLN modulation; model1.py:706, :711, :288)
__global__ void k_time_bwd_elem(const float* __restrict__ dtemb_atoms, const float* __restrict__ dstemb,
                                const float* __restrict__ temb, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = dtemb_atoms[i] + dstemb[i] * silu_grad(temb[i]);
}

// PILOT head backward for the per-molecule scalars (model1.py:354-359), one block per molecule:
//   pemb = sum_i w_i S_i,  w = sp / sum(sp),  sp = softplus(z)
//   dS_i = w_i dpemb;  dw_i = dpemb . S_i;  dsp_i = (dw_i - sum_k dw_k w_k) / sum(sp);  dz_i = dsp_i sigmoid(z_i)
__global__ __launch_bounds__(256) void k_pilot_combine_bwd(const float* __restrict__ dpemb, const float* __restrict__ stacked,
                                                           const float* __restrict__ z, const float* __restrict__ wn,
                                                           float* __restrict__ dstacked, float* __restrict__ dz) {
  __shared__ float sdw[8];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float d = dpemb[(size_t)b * 256 + tid];
  for (int i = 0; i < 8; ++i) dstacked[(size_t)b * 2048 + i * 256 + tid] = wn[(size_t)b * 8 + i] * d;
  const f32x4 dv = *reinterpret_cast<const f32x4*>(dpemb + (size_t)b * 256 + lane * 4);
  for (int i = wave; i < 8; i += 4) {
    const f32x4 sv = *reinterpret_cast<const f32x4*>(stacked + (size_t)b * 2048 + i * 256 + lane * 4);
    const float s = wave_sum((dv[0] * sv[0] + dv[1] * sv[1]) + (dv[2] * sv[2] + dv[3] * sv[3]));
    if (lane == 0) sdw[i] = s;
  }
  __syncthreads();
  if (tid < 8) {
    float dot = 0.f, tot = 0.f;
    for (int k = 0; k < 8; ++k) { dot += sdw[k] * wn[(size_t)b * 8 + k]; tot += softplus(z[(size_t)b * 8 + k]); }
    const float zi = z[(size_t)b * 8 + tid];
    dz[(size_t)b * 8 + tid] = (sdw[tid] - dot) / tot * (1.0f / (1.0f + expf(-zi)));
  }
}

// dg[b, k] = sum_i dz[b, i] Wp2[i, k], times gelu'(g_pre)   (pareto_predictor.2 input gradient)
__global__ __launch_bounds__(256) void k_pilot_dg(const float* __restrict__ dz, const float* __restrict__ Wp2,
                                                  const float* __restrict__ gpre, float* __restrict__ dgpre) {
  const int b = blockIdx.x, k = threadIdx.x;
  float a = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) a += dz[(size_t)b * 8 + i] * Wp2[i * 256 + k];
  dgpre[(size_t)b * 256 + k] = a * gelu_exact_grad(gpre[(size_t)b * 256 + k]);
}

// dWp2[i, k] += sum_b dz[b, i] g[b, k];  dbp2[i] += sum_b dz[b, i]
__global__ __launch_bounds__(256) void k_pilot_dw2(const float* __restrict__ dz, const float* __restrict__ g, int B,
                                                   float* __restrict__ dW, float* __restrict__ db) {
  const int k = threadIdx.x;
  const int b0 = blockIdx.x * 64, b1 = min(B, b0 + 64);
  float acc[8], bs = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  for (int b = b0; b < b1; ++b) {
    const float gv = g[(size_t)b * 256 + k];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += dz[(size_t)b * 8 + i] * gv;
    if (k < 8) bs += dz[(size_t)b * 8 + k];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) atomicAdd(dW + i * 256 + k, acc[i]);
  if (k < 8) atomicAdd(db + k, bs);
}

// Property-encoder backward (model1.py:341-345), one block per molecule, encoder e = blockIdx.y:
//   S_e = W2_e gelu(w0_e p_e + b0_e) + b2_e
//   dhid[j] = sum_o dS_e[o] W2_e[o, j];  dpre = dhid gelu'(pre);  dw0 += dpre p_e;  db0 += dpre
// dW2_e / db2_e are done by the GEMM / column-sum helpers.
struct PilotEncGrads { const float* W2[8]; float* dw0[8]; float* db0[8]; };
__global__ __launch_bounds__(64) void k_pilot_enc_bwd(const float* __restrict__ dstacked, const float* __restrict__ pre,
                                                      const float* __restrict__ props, PilotEncGrads p) {
  const int b = blockIdx.x, e = blockIdx.y, j = threadIdx.x;  // 64 threads = 64 hidden units
  const float* W2 = p.W2[e];
  float a = 0.f;
  for (int o = 0; o < 256; ++o) a += dstacked[(size_t)b * 2048 + e * 256 + o] * W2[o * 64 + j];
  const float dpre = a * gelu_exact_grad(pre[(size_t)b * 512 + e * 64 + j]);
  atomicAdd(p.dw0[e] + j, dpre * props[(size_t)b * LD_P + e]);
  atomicAdd(p.db0[e] + j, dpre);
}
