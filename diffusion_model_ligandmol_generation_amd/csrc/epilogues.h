// Model-specific row-wise epilogues for gemm_rows_kernel (see gemm.h for the register map).
#pragma once
#include "gemm.h"

// Input embedding, model1.py:709-718:
//   h = GELU(LN(x Wa^T + ba)) + GELU(LN(pos Wp^T + bp)) + t_emb[mol] (+ prop_emb[mol])
// The accumulators hold x Wa^T; the 3-wide position projection is done here on the VALU.
struct EpiEmbed {
  int M;
  const float* ba; const float* ga; const float* bea;
  const float* pos;            // [M,3]
  const float* Wp;             // [256,3]
  const float* bp; const float* gp; const float* bep;
  const float* temb; int temb_stride;   // [B or 1, 256]; stride 0 = one row for all
  const float* pemb; int pemb_stride;   // nullable
  const int32_t* atom_mol;
  float* H;
  float* ua_out;  // nullable: x Wa^T + ba (pre-LayerNorm), kept for backward
  __device__ __forceinline__ void operator()(f32x16 (&acc)[8], int row0, int by, int lane) const {
    const int c0 = lane & 31;
#pragma unroll 1
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + G_ROW(r, lane);
      const int rr = row < M ? row : M - 1;
      const float p0 = pos[(size_t)rr * 3 + 0], p1 = pos[(size_t)rr * 3 + 1], p2 = pos[(size_t)rr * 3 + 2];
      float v[8], u[8], s = 0.f, s2 = 0.f;
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) {
        const int col = ct * 32 + c0;
        v[ct] = acc[ct][r] + ba[col];
        // nn.Linear on a 3-vector: sum in k order, then bias
        u[ct] = (p0 * Wp[col * 3 + 0] + p1 * Wp[col * 3 + 1]) + p2 * Wp[col * 3 + 2] + bp[col];
        s += v[ct]; s2 += u[ct];
      }
      const float m1 = half_wave_sum(s) * (1.0f / 256.0f);
      const float m2 = half_wave_sum(s2) * (1.0f / 256.0f);
      float q1 = 0.f, q2 = 0.f;
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) {
        const float d1 = v[ct] - m1, d2 = u[ct] - m2;
        q1 += d1 * d1; q2 += d2 * d2;
      }
      const float r1 = 1.0f / sqrtf(half_wave_sum(q1) * (1.0f / 256.0f) + 1e-5f);
      const float r2 = 1.0f / sqrtf(half_wave_sum(q2) * (1.0f / 256.0f) + 1e-5f);
      if (row < M) {
        const int mol = atom_mol[row];
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
          const int col = ct * 32 + c0;
          const float y1 = gelu_exact((v[ct] - m1) * r1 * ga[col] + bea[col]);
          const float y2 = gelu_exact((u[ct] - m2) * r2 * gp[col] + bep[col]);
          float h = (y1 + y2) + temb[(size_t)mol * temb_stride + col];
          if (pemb != nullptr) h += pemb[(size_t)mol * pemb_stride + col];
          H[(size_t)row * 256 + col] = h;
          if (ua_out != nullptr) ua_out[(size_t)row * 256 + col] = v[ct];
        }
      }
    }
  }
};

// adaLN-gated residual, model1.py:295 / :300:  x = resid + gate[mol] * (acc + bias),
// optionally followed by the NEXT modulated pre-norm (model1.py:293 / :298)
//   xn = LN(x; eps 1e-6, no affine) * (1 + scale[mol]) + shift[mol].
// DROP (column-split form only): acc + bias is multiplied by the dropout mask of site `drop` first
// (model1.py:272: Dropout on the FFN output); proj_out then keeps the masked value.
template <bool LNMOD, bool DROP = false>
struct EpiGate {
  int M;
  const float* bias;
  const float* resid;                       // [M,256]
  const float* mod; int mod_stride; int gate_off;     // adaLN params of this layer
  const float* mod2; int shift_off; int scale_off;    // params feeding the next pre-norm
  const int32_t* atom_mol;
  float* X;                                 // [M,256]
  float* XN;                                // [M,256] (LNMOD)
  float* proj_out;                          // nullable: acc + bias (kept for backward)
  DropSite drop;                            // used with DROP only
  __device__ __forceinline__ void operator()(f32x16 (&acc)[8], int row0, int by, int lane) const {
    static_assert(!DROP, "dropout epilogue: column-split kernels only");
    const int c0 = lane & 31;
    float bv[8];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) bv[ct] = bias[ct * 32 + c0];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + G_ROW(r, lane);
      const int rr = row < M ? row : M - 1;
      const int mol = atom_mol[rr];
      const float* g = mod + (size_t)mol * mod_stride + gate_off;
      float v[8], s = 0.f;
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) {
        const int col = ct * 32 + c0;
        const float pr = acc[ct][r] + bv[ct];
        v[ct] = resid[(size_t)rr * 256 + col] + g[col] * pr;
        s += v[ct];
        if (proj_out != nullptr && row < M) proj_out[(size_t)row * 256 + col] = pr;
      }
      float mean = 0.f, rstd = 0.f;
      if (LNMOD) {
        mean = half_wave_sum(s) * (1.0f / 256.0f);
        float q = 0.f;
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) { const float d = v[ct] - mean; q += d * d; }
        rstd = 1.0f / sqrtf(half_wave_sum(q) * (1.0f / 256.0f) + 1e-6f);
      }
      if (row < M) {
        const float* m2 = mod2 + (size_t)mol * mod_stride;
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
          const int col = ct * 32 + c0;
          X[(size_t)row * 256 + col] = v[ct];
          if (LNMOD)
            XN[(size_t)row * 256 + col] =
                ((v[ct] - mean) * rstd) * (1.0f + m2[scale_off + col]) + m2[shift_off + col];
        }
      }
    }
  }
  // Column-split form.  Everything that does not depend on the accumulators is loaded up
  // front in one batch (residual rows, molecule ids, and - when all molecules share one
  // timestep, mod_stride == 0 - the gate / scale / shift vectors): a load left behind a
  // store through a possibly aliasing pointer (X may BE resid) cannot be hoisted by the
  // compiler and would cost one memory latency per row.
  template <bool SAVE, class MB>
  __device__ __forceinline__ void cs_body(f32x16 (&acc_t)[4], int row0, int by, int half, int lane, const MB& mb) const {
    float acc[4][16];   // plain scalars: element updates on the 16-register MFMA tuples would copy whole tuples
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][r] = acc_t[ct][r];
    const int c0 = lane & 31;
    const int cb = half * 128 + c0;
    const bool shared_mod = mod_stride == 0;
    float bv[4], g0[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) { bv[ct] = bias[cb + ct * 32]; g0[ct] = mod[gate_off + cb + ct * 32]; }
    // phase A (no stores unless SAVE): x = resid + gate * (acc + bias) in place, half-row statistics
    float mean[16], q2[16];
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      float rs[4][8], gt[4][8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = row0 + G_ROW(hb * 8 + i, lane);
        const int rr = row < M ? row : M - 1;
        const size_t mo = shared_mod ? 0 : (size_t)atom_mol[rr] * mod_stride;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          rs[ct][i] = resid[(size_t)rr * 256 + cb + ct * 32];
          gt[ct][i] = shared_mod ? g0[ct] : mod[mo + gate_off + cb + ct * 32];
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = hb * 8 + i;
        const int row = row0 + G_ROW(r, lane);
        float s = 0.f;
        f32x4 ms = {1.f, 1.f, 1.f, 1.f};
        if (DROP) ms = drop_scale4(drop, drop_row_quad(drop, row, cb, 256));
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const float pr = DROP ? (acc[ct][r] + bv[ct]) * ms[ct] : acc[ct][r] + bv[ct];
          if (SAVE && row < M) proj_out[(size_t)row * 256 + cb + ct * 32] = pr;
          acc[ct][r] = rs[ct][i] + gt[ct][i] * pr;
          s += acc[ct][r];
        }
        if (LNMOD) {
          mean[r] = half_wave_sum(s) * (1.0f / 128.0f);
          float q = 0.f;
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) { const float d = acc[ct][r] - mean[r]; q += d * d; }
          q2[r] = half_wave_sum(q);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (LNMOD) {
      mb.put(0, mean); mb.put(1, q2);
      __syncthreads();
      float pmean[16], pq2[16];
      mb.get_partner(0, pmean); mb.get_partner(1, pq2);
#pragma unroll
      for (int r = 0; r < 16; ++r) {   // mean[] <- row mean, q2[] <- 1/std
        const float dm = mean[r] - pmean[r];
        const float var = (q2[r] + pq2[r] + 64.0f * dm * dm) * (1.0f / 256.0f);
        mean[r] = 0.5f * (mean[r] + pmean[r]);
        q2[r] = 1.0f / sqrtf(var + 1e-6f);
      }
    }
    // phase B: stores; per-molecule scale / shift rows are fetched eight rows at a time
    float sc0[4], sh0[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      sc0[ct] = LNMOD ? mod2[scale_off + cb + ct * 32] : 0.f;
      sh0[ct] = LNMOD ? mod2[shift_off + cb + ct * 32] : 0.f;
    }
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      float sct[4][8], sht[4][8];
      if (LNMOD && !shared_mod) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int row = row0 + G_ROW(hb * 8 + i, lane);
          const int rr = row < M ? row : M - 1;
          const float* m2 = mod2 + (size_t)atom_mol[rr] * mod_stride + cb;
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) { sct[ct][i] = m2[scale_off + ct * 32]; sht[ct][i] = m2[shift_off + ct * 32]; }
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = hb * 8 + i;
        const int row = row0 + G_ROW(r, lane);
        if (row < M) {
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) {
            const int col = cb + ct * 32;
            X[(size_t)row * 256 + col] = acc[ct][r];
            if (LNMOD) {
              const float sc = shared_mod ? sc0[ct] : sct[ct][i];
              const float sh = shared_mod ? sh0[ct] : sht[ct][i];
              XN[(size_t)row * 256 + col] = ((acc[ct][r] - mean[r]) * q2[r]) * (1.0f + sc) + sh;
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  template <class MB>
  __device__ __forceinline__ void cs(f32x16 (&acc)[4], int row0, int by, int half, int lane, const MB& mb) const {
    if (proj_out != nullptr) cs_body<true>(acc, row0, by, half, lane, mb);
    else cs_body<false>(acc, row0, by, half, lane, mb);
  }
};

// First edge-MLP layer split by operand, model1.py:96-97 (Linear 544->256, SiLU):
//   W1 [h_src | h_dst | rbf] + b1 = Pa[src] + Pb[dst] + Wc rbf + b1
// with Pa|Pb = h [W1a|W1b]^T computed once per NODE (P is [N,512]); the accumulators hold
// Wc rbf (K = 32).  Stores SiLU(.) as the A operand of the second edge linear.
struct EpiEdgePre {
  int M;  // edges
  const float* b1;
  const float* P;  // [N,512]
  const int32_t* src; const int32_t* dst;
  float* A1;       // [E,256]
  float* pre_out;  // nullable: pre-activation
  __device__ __forceinline__ void operator()(f32x16 (&acc)[8], int row0, int by, int lane) const {
    const int c0 = lane & 31;
    float bv[8];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) bv[ct] = b1[ct * 32 + c0];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + G_ROW(r, lane);
      if (row < M) {
        const float* pa = P + (size_t)src[row] * 512;
        const float* pb = P + (size_t)dst[row] * 512 + 256;
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
          const int col = ct * 32 + c0;
          const float v = (pa[col] + pb[col]) + (acc[ct][r] + bv[ct]);
          A1[(size_t)row * 256 + col] = silu(v);
          if (pre_out != nullptr) pre_out[(size_t)row * 256 + col] = v;
        }
      }
    }
  }
  // Column-split form: all row indices, then all gathered P values, are loaded in two
  // batches before the first store (see EpiGate::cs for why).
  template <bool SAVE, class MB>
  __device__ __forceinline__ void cs_body(f32x16 (&acc)[4], int row0, int by, int half, int lane, const MB&) const {
    const int cb = half * 128 + (lane & 31);
    int is[16], id[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + G_ROW(r, lane);
      const int rr = row < M ? row : M - 1;
      is[r] = src[rr]; id[r] = dst[rr];
    }
    float bv[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) bv[ct] = b1[cb + ct * 32];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float* pa = P + (size_t)is[r] * 512 + cb;
      const float* pb = P + (size_t)id[r] * 512 + 256 + cb;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[ct][r] = (pa[ct * 32] + pb[ct * 32]) + (acc[ct][r] + bv[ct]);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + G_ROW(r, lane);
      if (row < M) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const int col = cb + ct * 32;
          A1[(size_t)row * 256 + col] = silu(acc[ct][r]);
          if (SAVE) pre_out[(size_t)row * 256 + col] = acc[ct][r];
        }
      }
    }
  }
  template <class MB>
  __device__ __forceinline__ void cs(f32x16 (&acc)[4], int row0, int by, int half, int lane, const MB& mb) const {
    if (pre_out != nullptr) cs_body<true>(acc, row0, by, half, lane, mb);
    else cs_body<false>(acc, row0, by, half, lane, mb);
  }
};

// Coordinate gate, model1.py:79-84 / :106-107 (NT = 4, 128 hidden units):
//   cw = tanh(w2 . SiLU(acc + b1) + b2);  wrel = rel * cw
struct EpiCoord {
  int M;
  const float* b1;   // [128]
  const float* w2;   // [128]
  const float* b2;   // [1]
  const float* rel;  // [E,4] (xyz + pad)
  float* wrel;       // [E,4]
  float* hid_out;    // nullable: [E,128] pre-activation
  float* cw_out;     // nullable: [E]
  __device__ __forceinline__ void operator()(f32x16 (&acc)[4], int row0, int by, int lane) const {
    if (hid_out != nullptr) body<true>(acc, row0, lane);
    else body<false>(acc, row0, lane);
  }
  template <bool SAVE>
  __device__ __forceinline__ void body(f32x16 (&acc)[4], int row0, int lane) const {
    const int c0 = lane & 31;
    float bv[4], wv[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) { bv[ct] = b1[ct * 32 + c0]; wv[ct] = w2[ct * 32 + c0]; }
    const float bias2 = b2[0];
    float rl[16];   // rel of this lane's coordinate, loaded ahead of the first store
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + G_ROW(r, lane);
      const int rr = row < M ? row : M - 1;
      rl[r] = c0 < 3 ? rel[(size_t)rr * 4 + c0] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + G_ROW(r, lane);
      float s = 0.f;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const float u = acc[ct][r] + bv[ct];
        s += silu(u) * wv[ct];
        if (SAVE && row < M) hid_out[(size_t)row * 128 + ct * 32 + c0] = u;
      }
      const float cw = tanhf(half_wave_sum(s) + bias2);
      if (row < M && c0 < 3) wrel[(size_t)row * 4 + c0] = rl[r] * cw;
      if (SAVE && row < M && c0 == 3) cw_out[row] = cw;
    }
  }
};
