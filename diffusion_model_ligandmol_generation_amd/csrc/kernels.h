// Non-GEMM kernels of the score-network forward (gather / scatter / normalisation /
// attention).  All are HBM- or latency-bound VALU kernels; one wave handles one row,
// one destination node or one (molecule, head) so that every global access is a
// contiguous 128-B..1-KiB segment.
#pragma once
#include "weights.h"
#include "common.h"

// accumulator row of register r in lane half h of a 32x32 MFMA tile
#define G_ROW_ATT(r, h) (((r) & 3) + 8 * ((r) >> 2) + 4 * (h))

// Fourier features, model1.py:686-697.  freqs[128] is computed on the host with the
// reference's own expression; t is the per-molecule value or one device-side step index.
__global__ void k_fourier(const float* __restrict__ t, const int32_t* __restrict__ t_index,
                          const float* __restrict__ freqs, float* __restrict__ out) {
  const int b = blockIdx.x, k = threadIdx.x;  // 128 threads
  const float tv = t_index != nullptr ? (float)(*t_index) : t[b];
  const float e = tv * freqs[k];
  out[(size_t)b * 256 + k] = sinf(e);
  out[(size_t)b * 256 + 128 + k] = cosf(e);
}

// Linear layer for a handful of rows (the time path when every molecule shares one step:
// M = 1): one wave per output feature, k split over the lanes.  A 128-row MFMA tile would
// run one workgroup for the whole layer; this spreads the weight read over the chip.
//   C[m, o] = act(A[m,:] . W[o,:] + b[o]),  optional C2 = SiLU(C).  blockIdx.y picks a set.
struct SmallLinearSets { const float* W[LD_MAXL]; const float* b[LD_MAXL]; float* C[LD_MAXL]; };
template <int ACT, bool DUAL_SILU>
__global__ __launch_bounds__(256) void k_small_linear(const float* __restrict__ A, SmallLinearSets p,
                                                      float* __restrict__ C2, int M, int K, int Nout) {
  const int lane = threadIdx.x & 63;
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= Nout) return;
  const float* W = p.W[blockIdx.y] + (size_t)o * K;
  for (int m = 0; m < M; ++m) {
    float s = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(W + k);
      const f32x4 a = *reinterpret_cast<const f32x4*>(A + (size_t)m * K + k);
      s += (a[0] * w[0] + a[1] * w[1]) + (a[2] * w[2] + a[3] * w[3]);
    }
    s = wave_sum(s);
    if (lane == 0) {
      float v = s + p.b[blockIdx.y][o];
      if (ACT == 1) v = gelu_exact(v);
      if (ACT == 2) v = silu(v);
      p.C[blockIdx.y][(size_t)m * Nout + o] = v;
      if (DUAL_SILU) C2[(size_t)m * Nout + o] = silu(v);
    }
  }
}

// Edge geometry + radial MLP, model1.py:88-93, 32 CSR edges per wave:
//   rel = pos[dst] - pos[src];  d = ||rel + 1e-8||;  rbf = W2 SiLU(w0 d + b0) + b2
// The 32 -> 32 layer is a 32x32x32 product per wave on the matrix cores
// (16 x v_mfma_f32_32x32x2_f32): lane (i, h) owns edge i and builds the hidden units k = 2s + h
// in registers as the A operand; the B operand is W2[n = lane & 31][2s + h].
__global__ __launch_bounds__(256) void k_edge_geom(
    const float* __restrict__ pos, const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
    const float* __restrict__ w0, const float* __restrict__ b0, const float* __restrict__ W2,
    const float* __restrict__ b2, float* __restrict__ rel, float* __restrict__ rbf,
    float* __restrict__ dist_out, float* __restrict__ r1pre_out, int E) {
  const int lane = threadIdx.x & 63, fi = lane & 31, h = lane >> 5;
  const int e0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 32;
  if (e0 >= E) return;
  const int e = e0 + fi;
  const int ec = e < E ? e : E - 1;
  const int sa = src[ec], da = dst[ec];
  const float rx = pos[(size_t)da * 3 + 0] - pos[(size_t)sa * 3 + 0];
  const float ry = pos[(size_t)da * 3 + 1] - pos[(size_t)sa * 3 + 1];
  const float rz = pos[(size_t)da * 3 + 2] - pos[(size_t)sa * 3 + 2];
  const float ex = rx + 1e-8f, ey = ry + 1e-8f, ez = rz + 1e-8f;
  const float dist = sqrtf(ex * ex + ey * ey + ez * ez);
  if (h == 0 && e < E) {
    const f32x4 rv = {rx, ry, rz, 0.f};
    *reinterpret_cast<f32x4*>(rel + (size_t)e * 4) = rv;
    if (dist_out != nullptr) dist_out[e] = dist;
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float* w2row = W2 + fi * 32 + h;
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int k = 2 * s + h;
    const float pre = dist * w0[k] + b0[k];
    if (r1pre_out != nullptr && e < E) r1pre_out[(size_t)e * 32 + k] = pre;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(silu(pre), w2row[2 * s], acc, 0, 0, 0);
  }
  const float bias = b2[fi];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = e0 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (row < E) rbf[(size_t)row * 32 + fi] = acc[r] + bias;
  }
}

// Position / time part of the input embedding, model1.py:710-718 (one wave per atom):
//   R = GELU(LN(pos Wp^T + bp)) + t_emb[mol] (+ prop_emb[mol])
// The atom-feature GEMM adds its own LayerNorm + GELU branch on top of R in its epilogue.
// With xpad != NULL the same launch also zero-extends the 119-wide atom features and the atom-embedding
// weight rows to 128 floats (512-B rows: the form the LDS-DMA GEMM needs, 16-B aligned rows and K a
// multiple of 16): the wave of atom `row` writes that atom's row of xpad, and 64 extra workgroups
// (blockIdx.x >= ceil(N / 4)) write the 256 rows of wpad.
__global__ __launch_bounds__(256) void k_pos_embed(const float* __restrict__ pos, const float* __restrict__ Wp,
                                                   const float* __restrict__ bp, const float* __restrict__ gp,
                                                   const float* __restrict__ bep, const float* __restrict__ temb,
                                                   int temb_stride, const float* __restrict__ pemb, int pemb_stride,
                                                   const int32_t* __restrict__ atom_mol, float* __restrict__ R, int N,
                                                   int pemb_mols,   // molecules [0, pemb_mols) receive pemb
                                                   const float* __restrict__ x = nullptr, const float* __restrict__ Wa = nullptr,
                                                   float* __restrict__ xpad = nullptr, float* __restrict__ wpad = nullptr) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (xpad != nullptr && lane < 32) {
    const int nb_atoms = (N + 3) >> 2;
    const float* src = nullptr;
    float* dst = nullptr;
    if (row < N) { src = x + (size_t)row * LD_A; dst = xpad + (size_t)row * 128; }
    else if ((int)blockIdx.x >= nb_atoms && row - nb_atoms * 4 < 256) {
      src = Wa + (size_t)(row - nb_atoms * 4) * LD_A; dst = wpad + (size_t)(row - nb_atoms * 4) * 128;
    }
    if (src != nullptr) {
      const int q = lane * 4;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (q + e) < LD_A ? src[q + e] : 0.f;
      *reinterpret_cast<f32x4*>(dst + q) = v;
    }
  }
  if (row >= N) return;
  const float p0 = pos[(size_t)row * 3 + 0], p1 = pos[(size_t)row * 3 + 1], p2 = pos[(size_t)row * 3 + 2];
  const int mol = atom_mol[row];
  f32x4 u;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int col = lane * 4 + i;
    u[i] = (p0 * Wp[col * 3 + 0] + p1 * Wp[col * 3 + 1]) + p2 * Wp[col * 3 + 2] + bp[col];
  }
  const float mean = wave_sum((u[0] + u[1]) + (u[2] + u[3])) * (1.0f / 256.0f);
  const f32x4 d = u - mean;
  const float var = wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.0f / 256.0f);
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
  const f32x4 g = *reinterpret_cast<const f32x4*>(gp + lane * 4);
  const f32x4 b = *reinterpret_cast<const f32x4*>(bep + lane * 4);
  f32x4 t = *reinterpret_cast<const f32x4*>(temb + (size_t)mol * temb_stride + lane * 4);
  f32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = gelu_exact(d[i] * rstd * g[i] + b[i]) + t[i];
  if (pemb != nullptr && mol < pemb_mols) o += *reinterpret_cast<const f32x4*>(pemb + (size_t)mol * pemb_stride + lane * 4);
  *reinterpret_cast<f32x4*>(R + (size_t)row * 256 + lane * 4) = o;
}

// Message aggregation + coordinate scatter, model1.py:100 / :110-112 / :724-725.
// One wave per destination atom.  The atom's incoming messages are CONTIGUOUS rows of
// msg (CSR order), so the sum streams 1-KiB rows, 16 B per lane, in the fixed edge order
// of a sequential index_add_ (bit-reproducible, no atomics).
//   agg[i]      = sum_{e in in(i)} msg[e]
//   dpos[i]     = sum_{e in in(i)} wrel[e] - sum_{e in out(i)} wrel[e]
//   pos_upd[i] += 0.1 dpos[i];  pos_cur[i] = pos[i] + pos_upd[i]
// Two atoms per wave (one per 32-lane half, 32 B of every row per lane) and all four CSR
// bounds loaded up front: an atom has ~2 incoming edges, so the kernel is a chain of three
// dependent memory latencies and its rate is set by how many of them are in flight.
// PARTS: `msg` holds the per-slab partial sums of the fused edge kernel's REDUCE form (edge_chain.h) instead of
// message rows: atom i adds rows i + slab(e0) .. i + slab(e1 - 1), in that order, where slab(e) counts 32-edge slabs
// from the first edge of the atom's molecule: mol_slab0[m] + ((e - mol_e0[m]) >> 5).
template <bool PARTS = false>
__global__ __launch_bounds__(256) void k_aggregate(
    const float* __restrict__ msg, const float* __restrict__ wrel,
    const int32_t* __restrict__ dst_ptr, const int32_t* __restrict__ src_ptr,
    const int32_t* __restrict__ src_slot, const float* __restrict__ pos,
    float* __restrict__ pos_upd, float* __restrict__ pos_cur, float* __restrict__ agg,
    float* __restrict__ dpos_out, int N, int first_layer, const int32_t* __restrict__ atom_mol = nullptr,
    const int32_t* __restrict__ mol_e0 = nullptr, const int32_t* __restrict__ mol_slab0 = nullptr) {
  const int l = threadIdx.x & 31;
  const int node = blockIdx.x * 8 + (threadIdx.x >> 5);
  if (node >= N) return;
  const int e0 = dst_ptr[node], e1 = dst_ptr[node + 1];
  const int s0 = src_ptr[node], s1 = src_ptr[node + 1];
  const int c = l & 3, g = l >> 2;
  // coordinate part first: its loads (wrel of incoming edges, slots of outgoing edges) go
  // out together with the first message rows
  float d = 0.f;
  int slot[2];
  const int k0 = s0 + g, k1 = s0 + g + 8;
  slot[0] = k0 < s1 ? src_slot[k0] : -1;
  slot[1] = k1 < s1 ? src_slot[k1] : -1;
  for (int k = e0 + g; k < e1; k += 8) d += wrel[(size_t)k * 4 + c];
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
  const float* mrow = msg + l * 4;
  int e = e0;
  int e1r = e1;   // the row range [e, e1r) to add
  if (PARTS && e1 > e0) {
    const int m = atom_mol[node];
    const int me0 = mol_e0[m], base = node + mol_slab0[m];
    e = base + ((e0 - me0) >> 5);
    e1r = base + ((e1 - 1 - me0) >> 5) + 1;
  }
  for (; e + 2 <= e1r; e += 2) {  // 2 rows (4 loads of 16 B) in flight per half-wave
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(mrow + (size_t)(e + 0) * 256);
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(mrow + (size_t)(e + 0) * 256 + 128);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(mrow + (size_t)(e + 1) * 256);
    const f32x4 w1 = *reinterpret_cast<const f32x4*>(mrow + (size_t)(e + 1) * 256 + 128);
    a0 = (a0 + v0) + v1;
    a1 = (a1 + w0) + w1;
  }
  if (e < e1r) {
    a0 += *reinterpret_cast<const f32x4*>(mrow + (size_t)e * 256);
    a1 += *reinterpret_cast<const f32x4*>(mrow + (size_t)e * 256 + 128);
  }
  *reinterpret_cast<f32x4*>(agg + (size_t)node * 256 + l * 4) = a0;
  *reinterpret_cast<f32x4*>(agg + (size_t)node * 256 + 128 + l * 4) = a1;

  if (slot[0] >= 0) d -= wrel[(size_t)slot[0] * 4 + c];
  if (slot[1] >= 0) d -= wrel[(size_t)slot[1] * 4 + c];
  for (int k = s0 + g + 16; k < s1; k += 8) d -= wrel[(size_t)src_slot[k] * 4 + c];
  d += __shfl_xor(d, 4);
  d += __shfl_xor(d, 8);
  d += __shfl_xor(d, 16);
  if (l < 3) {
    const size_t o = (size_t)node * 3 + l;
    if (dpos_out != nullptr) dpos_out[o] = d;
    const float pu = (first_layer ? 0.f : pos_upd[o]) + d * 0.1f;
    pos_upd[o] = pu;
    pos_cur[o] = pos[o] + pu;
  }
}

// Slab tables of the fused edge kernel's REDUCE form (edge_chain.h): the edges of molecule m are cut into slabs of 32
// counted from the molecule's first edge mol_e0[m] = dst_ptr[mol_ptr[m]]; mol_slab0 = exclusive prefix of the slab
// counts (mol_slab0[B] = slabs in use).  One workgroup; each thread scans a contiguous run of molecules.
__global__ __launch_bounds__(1024) void k_slab_scan(const int32_t* __restrict__ mol_ptr, const int32_t* __restrict__ dst_ptr,
                                                    int B, int32_t* __restrict__ mol_e0, int32_t* __restrict__ mol_slab0) {
  __shared__ int part[1024];
  const int t = threadIdx.x, per = (B + 1023) / 1024;
  const int m0 = t * per, m1 = min(B, m0 + per);
  int sum = 0;
  for (int m = m0; m < m1; ++m) sum += (dst_ptr[mol_ptr[m + 1]] - dst_ptr[mol_ptr[m]] + 31) >> 5;
  part[t] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {   // inclusive Hillis-Steele scan
    const int v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - sum;
  for (int m = m0; m < m1; ++m) {
    const int a = dst_ptr[mol_ptr[m]];
    mol_e0[m] = a;
    mol_slab0[m] = run;
    run += (dst_ptr[mol_ptr[m + 1]] - a + 31) >> 5;
  }
  if (t == 1023) { mol_e0[B] = dst_ptr[mol_ptr[B]]; mol_slab0[B] = part[1023]; }
}
// slab_e0 / slab_cnt [n_cap]: workgroups [0, ceil(B / 4)) fill the slabs of one molecule per wave, the rest clear the
// entries past the slabs in use.
__global__ __launch_bounds__(256) void k_slab_fill(const int32_t* __restrict__ mol_e0, const int32_t* __restrict__ mol_slab0,
                                                   int B, int n_cap, int32_t* __restrict__ slab_e0, int32_t* __restrict__ slab_cnt) {
  const int nb_mol = (B + 3) >> 2;
  if ((int)blockIdx.x < nb_mol) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= B) return;
    const int a = mol_e0[m], b = mol_e0[m + 1], s0 = mol_slab0[m];
    for (int k = lane; a + 32 * k < b; k += 64) {
      slab_e0[s0 + k] = a + 32 * k;
      slab_cnt[s0 + k] = min(32, b - (a + 32 * k));
    }
  } else {
    const int i = (blockIdx.x - nb_mol) * 256 + threadIdx.x;
    if (i < n_cap && i >= mol_slab0[B]) { slab_e0[i] = 0; slab_cnt[i] = 0; }
  }
}

// xn = LN(x; eps 1e-6, no affine) * (1 + scale[mol]) + shift[mol]   (model1.py:293)
__global__ __launch_bounds__(256) void k_ln_mod(const float* __restrict__ x, const float* __restrict__ mod,
                                                int mod_stride, int shift_off, int scale_off,
                                                const int32_t* __restrict__ atom_mol,
                                                float* __restrict__ xn, int N) {
  // four rows per wave, all loads first: a wave per row leaves one KiB in flight per wave and the launch
  // of ~30 000 short waves is then the cost
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
  if (row0 >= N) return;
  f32x4 v4[4], sc4[4], sh4[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int rc = row0 + q < N ? row0 + q : N - 1;
    v4[q] = *reinterpret_cast<const f32x4*>(x + (size_t)rc * 256 + lane * 4);
    const float* m = mod + (size_t)atom_mol[rc] * mod_stride;
    sc4[q] = *reinterpret_cast<const f32x4*>(m + scale_off + lane * 4);
    sh4[q] = *reinterpret_cast<const f32x4*>(m + shift_off + lane * 4);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = row0 + q;
    if (row >= N) break;
    const f32x4 v = v4[q];
    const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / 256.0f);
    const f32x4 dlt = v - mean;
    const float var = wave_sum((dlt[0] * dlt[0] + dlt[1] * dlt[1]) + (dlt[2] * dlt[2] + dlt[3] * dlt[3])) * (1.0f / 256.0f);
    const float rstd = 1.0f / sqrtf(var + 1e-6f);
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (dlt[i] * rstd) * (1.0f + sc4[q][i]) + sh4[q][i];
    *reinterpret_cast<f32x4*>(xn + (size_t)row * 256 + lane * 4) = o;
  }
}

// Per-molecule multi-head attention over the molecule's OWN atoms (8 heads x 32),
// model1.py:294 with nn.MultiheadAttention semantics.  Equal, for every real atom, to the
// reference's pad-to-max_atoms batch with key_padding_mask: masked keys get weight
// exactly 0 and padded query rows are dropped (model1.py:736-738).
// A group of G lanes serves one head (one query atom per lane, G = 32 or 64); the head's
// K and V rows sit in LDS and are read as broadcasts.  Softmax is the two-pass
// max / exp / sum form.  qkv is [N,768] = [q | k | v].
template <int G>
__global__ __launch_bounds__(256) void k_attention(const float* __restrict__ qkv,
                                                   const int32_t* __restrict__ mol_ptr,
                                                   float* __restrict__ out, float* __restrict__ lse_out,
                                                   int max_n) {
  constexpr int HPW = 64 / G;            // heads per wave
  constexpr int BPM = 8 / (4 * HPW);     // blocks per molecule
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mol = blockIdx.x / BPM, hb = blockIdx.x % BPM;
  const int a0 = mol_ptr[mol];
  const int n = mol_ptr[mol + 1] - a0;
  const int head0 = (hb * 4 + wave) * HPW;
  float* kv = smem + (size_t)wave * HPW * max_n * 64;
  for (int i = lane; i < n * HPW * 16; i += 64) {
    const int j = i / (HPW * 16), rem = i % (HPW * 16);
    const int hh = rem >> 4, part = rem & 15;
    const float* sp = qkv + (size_t)(a0 + j) * 768 + (part < 8 ? 256 : 512) + (head0 + hh) * 32 + (part & 7) * 4;
    *reinterpret_cast<f32x4*>(kv + ((size_t)hh * max_n + j) * 64 + part * 4) = *reinterpret_cast<const f32x4*>(sp);
  }
  __syncthreads();
  const int hh = lane / G;
  const float* kvh = kv + (size_t)hh * max_n * 64;
  const float scale = 0.17677669529663687f;  // 1/sqrt(32)
  for (int qi = lane % G; qi < n; qi += G) {
    f32x4 q[8];
    const float* qp = qkv + (size_t)(a0 + qi) * 768 + (head0 + hh) * 32;
#pragma unroll
    for (int d = 0; d < 8; ++d) q[d] = *reinterpret_cast<const f32x4*>(qp + d * 4);
    float m = -INFINITY;
    for (int j = 0; j < n; ++j) {
      const float* kp = kvh + (size_t)j * 64;
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        const f32x4 kk = *reinterpret_cast<const f32x4*>(kp + d * 4);
        s += q[d][0] * kk[0] + q[d][1] * kk[1] + q[d][2] * kk[2] + q[d][3] * kk[3];
      }
      m = fmaxf(m, s * scale);
    }
    float l = 0.f;
    f32x4 o[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < n; ++j) {
      const float* kp = kvh + (size_t)j * 64;
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        const f32x4 kk = *reinterpret_cast<const f32x4*>(kp + d * 4);
        s += q[d][0] * kk[0] + q[d][1] * kk[1] + q[d][2] * kk[2] + q[d][3] * kk[3];
      }
      const float p = expf(s * scale - m);
      l += p;
#pragma unroll
      for (int d = 0; d < 8; ++d) o[d] += p * *reinterpret_cast<const f32x4*>(kp + 32 + d * 4);
    }
    const float inv = 1.0f / l;
    float* op = out + (size_t)(a0 + qi) * 256 + (head0 + hh) * 32;
#pragma unroll
    for (int d = 0; d < 8; ++d) *reinterpret_cast<f32x4*>(op + d * 4) = o[d] * inv;
    if (lse_out != nullptr) lse_out[(size_t)(a0 + qi) * 8 + head0 + hh] = m + logf(l);
  }
}

// MFMA form of the same attention (fp32 v_mfma_f32_32x32x2_f32, exact fp32 dot products).
// One wave per (molecule, head, 32-query tile).  Scores are computed TRANSPOSED,
//   S^T[j, i] = sum_d K[j, d] (scale Q[i, d]),
// so a lane owns ONE query i (accumulator column) and holds 16 of every 32 keys in its
// registers: the softmax max / sum are in-lane plus one cross-half shuffle.  The
// probability tile is then fed straight back as the B operand of
//   O^T[d, i] = sum_j V[j, d] P^T[j, i]
// (accumulator register r of lane half h is key j = (r&3) + 8(r>>2) + 4h, and the A operand
// fetches V[j(r,h), d] to match), so P never leaves registers.  K and V of the head sit in
// LDS ([n][33] floats: conflict-free column reads); keys beyond n get weight 0, queries
// beyond n are not stored.  TK = number of 32-key tiles (n <= 32 TK).
template <int TK>
__global__ __launch_bounds__(256) void k_attention_mfma(const float* __restrict__ qkv,
                                                        const int32_t* __restrict__ mol_ptr,
                                                        float* __restrict__ out, float* __restrict__ lse_out,
                                                        int n_mols) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NP = TK * 32;          // padded key count
  constexpr int ITEMS = 8 * TK;        // (head, query tile) items per molecule
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int item = blockIdx.x * 4 + wave;
  const int mol = item / ITEMS, rem = item % ITEMS;
  const int head = rem / TK, qt = rem % TK;
  float* sk = smem + (size_t)wave * (2 * NP * 33);
  float* sv = sk + NP * 33;
  int a0 = 0, n = 0;
  if (mol < n_mols) { a0 = mol_ptr[mol]; n = mol_ptr[mol + 1] - a0; }
  const bool active = (mol < n_mols) && (qt * 32 < n);
  if (active) {
    for (int i = lane; i < NP * 8; i += 64) {   // stage K, V rows of this head (zeros beyond n)
      const int j = i >> 3, c = (i & 7) * 4;
      f32x4 kk = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
      if (j < n) {
        const float* r = qkv + (size_t)(a0 + j) * 768 + head * 32 + c;
        kk = *reinterpret_cast<const f32x4*>(r + 256);
        vv = *reinterpret_cast<const f32x4*>(r + 512);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { sk[j * 33 + c + e] = kk[e]; sv[j * 33 + c + e] = vv[e]; }
    }
  }
  __syncthreads();
  if (!active) return;
  const int fi = lane & 31, h = lane >> 5;
  const int qi = qt * 32 + fi;                    // this lane's query (both halves share it)
  const float scale = 0.17677669529663687f;       // 1/sqrt(32)
  float q[16];                                    // Q[qi, 2s + h] * scale, s = 0..15
  {
    const int qr = qi < n ? qi : n - 1;
    const float* qp = qkv + (size_t)(a0 + qr) * 768 + head * 32;
#pragma unroll
    for (int s = 0; s < 16; ++s) q[s] = qp[2 * s + h] * scale;
  }
  f32x16 sc[TK];
#pragma unroll
  for (int t = 0; t < TK; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[t][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s)    // A = K[t*32 + fi, 2s + h], B = q
      sc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(sk[(t * 32 + fi) * 33 + 2 * s + h], q[s], sc[t], 0, 0, 0);
  }
  // softmax over keys: lane (i, h) holds keys j = t*32 + G_ROW(r, lane)
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < TK; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = t * 32 + G_ROW_ATT(r, h);
      sc[t][r] = j < n ? sc[t][r] : -INFINITY;
      m = fmaxf(m, sc[t][r]);
    }
  m = fmaxf(m, __shfl_xor(m, 32));
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < TK; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __expf(sc[t][r] - m);       // exp(-inf) = 0 for padded keys
      sc[t][r] = p;
      l += p;
    }
  l += __shfl_xor(l, 32);
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
  for (int t = 0; t < TK; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r)   // A = V[j(r,h), d = fi], B = P^T[j(r,h), i]
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[(t * 32 + G_ROW_ATT(r, h)) * 33 + fi], sc[t][r], o, 0, 0, 0);
  if (qi < n) {
    const float inv = 1.0f / l;
    float* op = out + (size_t)(a0 + qi) * 256 + head * 32;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {   // registers 4g..4g+3 are d = 8g + 4h + (0..3)
      f32x4 v = {o[gq * 4 + 0] * inv, o[gq * 4 + 1] * inv, o[gq * 4 + 2] * inv, o[gq * 4 + 3] * inv};
      *reinterpret_cast<f32x4*>(op + gq * 8 + h * 4) = v;
    }
    if (lse_out != nullptr && h == 0) lse_out[(size_t)(a0 + qi) * 8 + head] = m + logf(l);
  }
}

// Register-only form of k_attention_mfma (same products, same ownership: one wave per molecule, head
// and 32-query tile), without LDS and without a barrier.  The k index of an MFMA step is free as long
// as both operands agree, so step s of lane half h takes head dimension d = 16 h + s: a lane's 16
// values of its K row (and of its Q row) are then CONTIGUOUS - four 16-B loads straight from qkv into
// the operand registers.  For O^T = V^T P^T lane (d, h) needs V[j(r, h), d] for its 16 keys: 16 dword
// loads, each a coalesced 128-B row segment per lane half.  Rows beyond n are clamped to the last atom
// (their scores are masked to -inf, their probabilities are exactly 0).
// DROP: the normalised probabilities are multiplied by the dropout mask of site `drop` before the
// product with V (nn.MultiheadAttention in train mode, model1.py:263-267); the row sum l and the
// saved log-sum-exp stay those of the unmasked softmax.
template <int TK, bool DROP = false>
__global__ __launch_bounds__(256, TK >= 3 ? 4 : 1) void k_attention_reg(const float* __restrict__ qkv,
                                                       const int32_t* __restrict__ mol_ptr,
                                                       float* __restrict__ out, float* __restrict__ lse_out,
                                                       int n_mols, DropSite drop = DropSite{}) {
  constexpr int ITEMS = 8 * TK;
  // the wave index, and with it the molecule, head, query tile and every tile bound below, is the same for all
  // lanes: kept in scalar registers, so the tile branches are scalar branches and the row addresses 32-bit offsets
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int item = blockIdx.x * 4 + wave;
  const int mol = item / ITEMS, rem = item % ITEMS;
  const int head = rem / TK, qt = rem % TK;
  if (mol >= n_mols) return;
  const int a0 = mol_ptr[mol], n = mol_ptr[mol + 1] - a0;
  if (qt * 32 >= n) return;
  const int fi = lane & 31, h = lane >> 5;
  const int qi = qt * 32 + fi;
  const float scale = 0.17677669529663687f;       // 1/sqrt(32)
  const float* base = qkv + (size_t)a0 * 768 + head * 32;
  float q[16];
  {
    const uint32_t qo = (uint32_t)(qi < n ? qi : n - 1) * 768u + 16u * h;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(base + qo + c * 4);
      q[c * 4 + 0] = v[0] * scale; q[c * 4 + 1] = v[1] * scale; q[c * 4 + 2] = v[2] * scale; q[c * 4 + 3] = v[3] * scale;
    }
  }
  // key tiles beyond the molecule (t * 32 >= n, the same for every lane of the wave) are skipped: their scores
  // are masked to -inf below and their probabilities are exactly 0.  The K rows of tile t + 1 are requested before
  // the products of tile t, so that only two tiles of operands are live at a time.
  f32x16 sc[TK];
  f32x4 kf[2][4];
  auto load_k = [&](int t, f32x4* dst) {
    const int kr = t * 32 + fi;
    const uint32_t ko = (uint32_t)(kr < n ? kr : n - 1) * 768u + 256u + 16u * h;
#pragma unroll
    for (int c = 0; c < 4; ++c) dst[c] = *reinterpret_cast<const f32x4*>(base + ko + c * 4);
  };
  load_k(0, kf[0]);
#pragma unroll
  for (int t = 0; t < TK; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[t][r] = 0.f;
    if (t * 32 < n) {
      if (t + 1 < TK && (t + 1) * 32 < n) load_k(t + 1, kf[(t + 1) & 1]);
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e)   // A = K[t*32 + fi, 16h + 4c + e], B = scale Q[qi, same d]
          sc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[t & 1][c][e], q[c * 4 + e], sc[t], 0, 0, 0);
    }
  }
  // V[j(r, h), d = fi]: tile 0 is fetched while the softmax runs, tile t + 1 while tile t multiplies
  float vf[2][16];
  auto load_v = [&](int t, float* dst) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int vr = t * 32 + G_ROW_ATT(r, h);
      dst[r] = base[(uint32_t)(vr < n ? vr : n - 1) * 768u + 512u + fi];
    }
  };
  load_v(0, vf[0]);
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < TK; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = t * 32 + G_ROW_ATT(r, h);
      sc[t][r] = j < n ? sc[t][r] : -INFINITY;
      m = fmaxf(m, sc[t][r]);
    }
  m = fmaxf(m, __shfl_xor(m, 32));
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < TK; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __expf(sc[t][r] - m);
      sc[t][r] = p;
      l += p;
    }
  l += __shfl_xor(l, 32);
  if (DROP) {   // registers 4g..4g+3 of tile t are the keys j = 32 t + 8 g + 4 h + (0..3): one mask quad
#pragma unroll
    for (int t = 0; t < TK; ++t)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const f32x4 ms = drop_scale4(drop, drop_att_quad(drop, a0 + qi, head, t * 32 + gq * 8 + h * 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) sc[t][gq * 4 + e] *= ms[e];
      }
  }
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
  for (int t = 0; t < TK; ++t)
    if (t * 32 < n) {
      if (t + 1 < TK && (t + 1) * 32 < n) load_v(t + 1, vf[(t + 1) & 1]);
#pragma unroll
      for (int r = 0; r < 16; ++r) o = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[t & 1][r], sc[t][r], o, 0, 0, 0);
    }
  if (qi < n) {
    const float inv = 1.0f / l;
    float* op = out + (size_t)(a0 + qi) * 256 + head * 32;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {   // registers 4g..4g+3 are d = 8g + 4h + (0..3)
      f32x4 v = {o[gq * 4 + 0] * inv, o[gq * 4 + 1] * inv, o[gq * 4 + 2] * inv, o[gq * 4 + 3] * inv};
      *reinterpret_cast<f32x4*>(op + gq * 8 + h * 4) = v;
    }
    if (lse_out != nullptr && h == 0) lse_out[(size_t)(a0 + qi) * 8 + head] = m + logf(l);
  }
}

// out[row, o] = y[row, off:off+256] . W[o, :] + b[o] (+ add[row, o]); final layers of
// the coordinate (3) and property (15) heads, model1.py:647 / :654 / :742.
template <int NOUT>
__global__ __launch_bounds__(256) void k_head_small(const float* __restrict__ y, int ldy, int off,
                                                    const float* __restrict__ W, const float* __restrict__ b,
                                                    const float* __restrict__ add, float* __restrict__ out, int M) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;   // four rows per wave, loads first
  if (row0 >= M) return;
  f32x4 v4[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    v4[q] = *reinterpret_cast<const f32x4*>(y + (size_t)(row0 + q < M ? row0 + q : M - 1) * ldy + off + lane * 4);
#pragma unroll
  for (int o = 0; o < NOUT; ++o) {
    const f32x4 w = *reinterpret_cast<const f32x4*>(W + (size_t)o * 256 + lane * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = row0 + q;
      const f32x4 v = v4[q];
      float s = wave_sum((v[0] * w[0] + v[1] * w[1]) + (v[2] * w[2] + v[3] * w[3]));
      if (lane == 0 && row < M) {
        s += b[o];
        if (add != nullptr) s += add[(size_t)row * NOUT + o];
        out[(size_t)row * NOUT + o] = s;
      }
    }
  }
}

// global_mean_pool(h, batch), model1.py:745: one wave per molecule.
__global__ __launch_bounds__(256) void k_mean_pool(const float* __restrict__ h, const int32_t* __restrict__ mol_ptr,
                                                   float* __restrict__ out, int B) {
  const int lane = threadIdx.x & 63;
  const int mol = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (mol >= B) return;
  const int a0 = mol_ptr[mol], a1 = mol_ptr[mol + 1];
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int a = a0; a < a1; ++a) s += *reinterpret_cast<const f32x4*>(h + (size_t)a * 256 + lane * 4);
  const float inv = 1.0f / (float)(a1 - a0 > 0 ? a1 - a0 : 1);
  *reinterpret_cast<f32x4*>(out + (size_t)mol * 256 + lane * 4) = s * inv;
}

// PILOT property encoders, model1.py:341-351: stacked[b, i, :] = W2_i GELU(w0_i p_i + b0_i) + b2_i
struct PilotEncWeights { const float* w0[8]; const float* b0[8]; const float* W2[8]; const float* b2[8]; };
__global__ __launch_bounds__(256) void k_pilot_encode(const float* __restrict__ props, PilotEncWeights w,
                                                      float* __restrict__ stacked, float* __restrict__ pre_out) {
  __shared__ float hid[8][64];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < 512; i += 256) {
    const int e = i >> 6, j = i & 63;
    const float pre = props[(size_t)b * LD_P + e] * w.w0[e][j] + w.b0[e][j];
    hid[e][j] = gelu_exact(pre);
    if (pre_out != nullptr) pre_out[(size_t)b * 512 + i] = pre;
  }
  __syncthreads();
#pragma unroll 1
  for (int e = 0; e < 8; ++e) {
    const float* W2 = w.W2[e] + (size_t)tid * 64;
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < 64; ++j) a += W2[j] * hid[e][j];
    stacked[(size_t)b * 2048 + e * 256 + tid] = a + w.b2[e][tid];
  }
}

// Pareto weights + weighted sum, model1.py:354-359:
//   w = softplus(W g + b);  w /= sum(w);  out = sum_i w_i stacked[b, i, :]
__global__ __launch_bounds__(256) void k_pilot_combine(const float* __restrict__ g, const float* __restrict__ W,
                                                       const float* __restrict__ bias,
                                                       const float* __restrict__ stacked,
                                                       float* __restrict__ out, float* __restrict__ z_out,
                                                       float* __restrict__ w_out) {
  __shared__ float wsm[8];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const f32x4 gv = *reinterpret_cast<const f32x4*>(g + (size_t)b * 256 + lane * 4);
  for (int o = wave; o < 8; o += 4) {
    const f32x4 wv = *reinterpret_cast<const f32x4*>(W + (size_t)o * 256 + lane * 4);
    const float s = wave_sum((gv[0] * wv[0] + gv[1] * wv[1]) + (gv[2] * wv[2] + gv[3] * wv[3]));
    if (lane == 0) {
      wsm[o] = softplus(s + bias[o]);
      if (z_out != nullptr) z_out[(size_t)b * 8 + o] = s + bias[o];
    }
  }
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int o = 0; o < 8; ++o) tot += wsm[o];
  float a = 0.f;
#pragma unroll
  for (int o = 0; o < 8; ++o) a += stacked[(size_t)b * 2048 + o * 256 + tid] * (wsm[o] / tot);
  out[(size_t)b * 256 + tid] = a;
  if (w_out != nullptr && tid < 8) w_out[(size_t)b * 8 + tid] = wsm[tid] / tot;
}
