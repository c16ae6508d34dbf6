// MolDiff consistency module, model1.py:453-561, batched.
//
// The reference runs its bond predictor once per ordered atom pair inside a Python double loop
// (model1.py:513-540).  Here all pairs of all molecules form one row set: the first linear layer
// (516 -> 256 on [h_i | h_j | 4 distance features]) is split into two per-ATOM projections
// P = h [Wi | Wj]^T plus a 4-term sum per pair, so the expensive part costs N rows, not sum n(n-1);
// the remaining layers are ordinary row GEMMs over the pairs.  Dropout (model1.py:390, :401) is
// not applied.
#include "forward.h"
#include "launch.h"
#include "weights.h"

// softmax over the 119 atom-type logits of a row -> probs zero-extended to 128 columns (the
// operand form of the valency GEMM) and the arg-max (first maximum, as torch.argmax).
__global__ __launch_bounds__(256) void k_atom_softmax(const float* __restrict__ logits, float* __restrict__ probs128,
                                                      int32_t* __restrict__ pred, int N) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const float* l = logits + (size_t)row * LD_A;
  const float a = l[lane];
  const float b = lane + 64 < LD_A ? l[lane + 64] : -INFINITY;
  float m = fmaxf(a, b);
  int idx = a >= b ? lane : lane + 64;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const float om = __shfl_xor(m, o);
    const int oi = __shfl_xor(idx, o);
    if (om > m || (om == m && oi < idx)) { m = om; idx = oi; }
  }
  const float ea = expf(a - m), eb = lane + 64 < LD_A ? expf(b - m) : 0.f;
  const float inv = 1.0f / wave_sum(ea + eb);
  probs128[(size_t)row * 128 + lane] = ea * inv;
  probs128[(size_t)row * 128 + 64 + lane] = eb * inv;
  if (lane == 0) pred[row] = idx;
}

// dst[r, 0:ld_dst] = [src[r, 0:cols], 0...]
__global__ __launch_bounds__(256) void k_pad_cols(const float* __restrict__ src, int rows, int cols, int ld_src,
                                                  float* __restrict__ dst, int ld_dst) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)rows * ld_dst) return;
  const int r = (int)(i / ld_dst), c = (int)(i % ld_dst);
  dst[i] = c < cols ? src[(size_t)r * ld_src + c] : 0.f;
}

// valency[row] = softplus(y[row, 0:128] . w + b)   (model1.py:412-413), one wave per row
__global__ __launch_bounds__(256) void k_valency_out(const float* __restrict__ y, const float* __restrict__ w,
                                                     const float* __restrict__ b, float* __restrict__ val, int N) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const float* r = y + (size_t)row * 128;
  const float s = wave_sum(r[lane] * w[lane] + r[lane + 64] * w[lane + 64]);
  if (lane == 0) val[row] = softplus(s + b[0]);
}

// model1.py:466-468: torch.min(valency_pred.squeeze() [N], max_allowed.unsqueeze(-1) [N,1]) broadcasts
// to [N,N]:  out[i, j] = min(valency[j], max_valences[pred[i]]).
__global__ __launch_bounds__(256) void k_valency_constrain(const float* __restrict__ val, const int32_t* __restrict__ pred,
                                                           const float* __restrict__ maxval, float* __restrict__ out,
                                                           long N) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= N * N) return;
  const long r = i / N, c = i % N;
  int p = pred[r];
  p = p < 0 ? 0 : (p > 118 ? 118 : p);
  out[i] = fminf(val[c], maxval[p]);
}

// adjacency [B, n_max, n_max, 5] = 0 with [1,0,0,0,0] on the diagonal of each molecule's real atoms
// (model1.py:510, :515-518).
__global__ __launch_bounds__(256) void k_adj_init(float* __restrict__ adj, const int32_t* __restrict__ mol_ptr,
                                                  int n_max, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int k = (int)(i % 5);
  const long cell = i / 5;
  const int c = (int)(cell % n_max);
  const long rc = cell / n_max;
  const int r = (int)(rc % n_max);
  const int mol = (int)(rc / n_max);
  const int n = mol_ptr[mol + 1] - mol_ptr[mol];
  adj[i] = (k == 0 && r == c && r < n) ? 1.0f : 0.0f;
}

// Pair p of molecule m (n atoms, first atom a0, first pair q0 = pair_ptr[m]): i = (p - q0) / (n - 1),
// jj = (p - q0) % (n - 1), j = jj + (jj >= i)   - the order of the double loop at model1.py:513-540.
__device__ __forceinline__ void pair_of(const int32_t* mol_ptr, const int64_t* pair_ptr, int n_mols, int64_t p,
                                        int& mol, int& i, int& j, int& a0) {
  int lo = 0, hi = n_mols - 1;
  while (lo < hi) {   // last molecule with pair_ptr[m] <= p
    const int mid = (lo + hi + 1) >> 1;
    if (pair_ptr[mid] <= p) lo = mid; else hi = mid - 1;
  }
  mol = lo;
  a0 = mol_ptr[mol];
  const int n = mol_ptr[mol + 1] - a0;
  const int64_t q = p - pair_ptr[mol];
  i = (int)(q / (n - 1));
  const int jj = (int)(q % (n - 1));
  j = jj + (jj >= i ? 1 : 0);
}

// First bond-predictor layer per pair, one wave per pair (model1.py:520-536, :396-398):
//   u = Pi[i] + Pj[j] + Wf f + b,  f = [d, L, d / L, d < 3],  y = LN(GELU(u))
// with d = |pos_i - pos_j| and L = radius[type_i] + radius[type_j] (model1.py:553-561).
__global__ __launch_bounds__(256) void k_bond_pre(const float* __restrict__ P, const float* __restrict__ pos,
                                                  const int32_t* __restrict__ pred, const float* __restrict__ radii,
                                                  const float* __restrict__ W0, const float* __restrict__ b0,
                                                  const float* __restrict__ g, const float* __restrict__ be,
                                                  const int32_t* __restrict__ mol_ptr, const int64_t* __restrict__ pair_ptr,
                                                  int n_mols, int64_t n_pairs, float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n_pairs) return;
  int mol, i, j, a0;
  pair_of(mol_ptr, pair_ptr, n_mols, p, mol, i, j, a0);
  const int ai = a0 + i, aj = a0 + j;
  const float dx = pos[(size_t)ai * 3 + 0] - pos[(size_t)aj * 3 + 0];
  const float dy = pos[(size_t)ai * 3 + 1] - pos[(size_t)aj * 3 + 1];
  const float dz = pos[(size_t)ai * 3 + 2] - pos[(size_t)aj * 3 + 2];
  const float d = sqrtf(dx * dx + dy * dy + dz * dz);
  const int ti = min(pred[ai], 118), tj = min(pred[aj], 118);
  const float L = radii[ti] + radii[tj];
  const float f2 = d / L, f3 = d < 3.0f ? 1.0f : 0.0f;
  const f32x4 pi = *reinterpret_cast<const f32x4*>(P + (size_t)ai * 512 + lane * 4);
  const f32x4 pj = *reinterpret_cast<const f32x4*>(P + (size_t)aj * 512 + 256 + lane * 4);
  const f32x4 bb = *reinterpret_cast<const f32x4*>(b0 + lane * 4);
  f32x4 u;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float* wf = W0 + (size_t)(lane * 4 + e) * 516 + 512;
    u[e] = gelu_exact((pi[e] + pj[e]) + (((d * wf[0] + L * wf[1]) + f2 * wf[2]) + f3 * wf[3]) + bb[e]);
  }
  const float mean = wave_sum((u[0] + u[1]) + (u[2] + u[3])) * (1.0f / 256.0f);
  const f32x4 c = u - mean;
  const float var = wave_sum((c[0] * c[0] + c[1] * c[1]) + (c[2] * c[2] + c[3] * c[3])) * (1.0f / 256.0f);
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
  const f32x4 gg = *reinterpret_cast<const f32x4*>(g + lane * 4);
  const f32x4 bt = *reinterpret_cast<const f32x4*>(be + lane * 4);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = c[e] * rstd * gg[e] + bt[e];
  *reinterpret_cast<f32x4*>(y + (size_t)p * 256 + lane * 4) = o;
}

// Last bond layer + softmax + scatter (model1.py:536-540), one wave per pair:
//   logits = y2[p] W6^T + b6 (5 outputs);  adjacency[mol, i, j, :] = softmax(logits)
__global__ __launch_bounds__(256) void k_bond_out(const float* __restrict__ y2, const float* __restrict__ W6,
                                                  const float* __restrict__ b6, const int32_t* __restrict__ mol_ptr,
                                                  const int64_t* __restrict__ pair_ptr, int n_mols, int64_t n_pairs,
                                                  int n_max, float* __restrict__ bond_logits, float* __restrict__ adj) {
  const int lane = threadIdx.x & 63;
  const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n_pairs) return;
  const float* r = y2 + (size_t)p * 128;
  const float v0 = r[lane], v1 = r[lane + 64];
  float lg[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) lg[k] = wave_sum(v0 * W6[k * 128 + lane] + v1 * W6[k * 128 + 64 + lane]) + b6[k];
  if (lane == 0) {
    int mol, i, j, a0;
    pair_of(mol_ptr, pair_ptr, n_mols, p, mol, i, j, a0);
    float m = lg[0];
#pragma unroll
    for (int k = 1; k < 5; ++k) m = fmaxf(m, lg[k]);
    float e[5], s = 0.f;
#pragma unroll
    for (int k = 0; k < 5; ++k) { e[k] = expf(lg[k] - m); s += e[k]; }
    float* a = adj + (((size_t)mol * n_max + i) * n_max + j) * 5;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      bond_logits[(size_t)p * 5 + k] = lg[k];
      a[k] = e[k] / s;
    }
  }
}

struct ConsLayout {
  size_t y1, y2, probs, wv, val, pred, P, yb1, yb2, total;
};
static void cons_layout(ConsLayout* L, int64_t N, int64_t NP) {
  size_t off = 0;
  auto take = [&](size_t nfloats) { size_t o = off; off += align_up(nfloats * 4, 256); return o; };
  L->y1 = take(N * 256); L->y2 = take(N * 128); L->probs = take(N * 128); L->wv = take(256 * 384);
  L->val = take(N); L->pred = take(N); L->P = take(N * 512);
  L->yb1 = take((size_t)NP * 256); L->yb2 = take((size_t)NP * 128);
  L->total = off;
}

extern "C" size_t ligdiff_consistency_workspace_bytes(int64_t n_atoms, int64_t n_pairs) {
  if (n_atoms <= 0 || n_pairs < 0) return 0;
  ConsLayout L;
  cons_layout(&L, n_atoms, n_pairs);
  return L.total;
}

// hidden -> 256 (GELU, LN) -> 128 (GELU): the trunk shared by the three predictors' tails
static void cons_trunk_tail(const float* const* w, int pred, const float* y1, int M, float* y2, hipStream_t s) {
  GemmCore g = core_blocks(M, 256, 128, w[W_CONS(pred, CONS_W3)], 256, 1, 128);
  EpiBiasAct<4, ACT_GELU, false> e;
  memset(&e, 0, sizeof e);
  e.M = M; e.n_valid = 128; e.col_block = 0; e.bias[0] = w[W_CONS(pred, CONS_B3)]; e.C = y2; e.ldc = 128; e.alpha = 1.0f;
  launch_gemm_plain<4>(g, y1, 256, e, 1, s);
}

// y[row, col] *= mask scale of dropout site d, in place ([rows, 128]): the Dropout between the trunk's
// second GELU and the last Linear of the atom-type / bond predictor (model1.py:390, :401).
__global__ void k_dropout_rows128(float* __restrict__ y, int64_t rows, DropSite d) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * 128) return;
  y[i] *= drop_row_scale(d, (int)(i >> 7), (int)(i & 127), 128);
}

extern "C" int ligdiff_consistency_train(const float* const* w, const ligdiff_graph* g, const float* h, const float* pos,
                                         const int64_t* pair_ptr, int64_t n_pairs, int32_t n_max, float* atom_logits,
                                         float* bond_logits, float* valency, float* adjacency, void* workspace,
                                         size_t workspace_bytes, void* stream, const ligdiff_dropout* dropout) {
  DropSite d_atom, d_bond;
  if (!make_drop_site(dropout, 0, LIGDIFF_DROP_CONS_ATOM, &d_atom) || !make_drop_site(dropout, 0, LIGDIFF_DROP_CONS_BOND, &d_bond))
    return LIGDIFF_ERR_ARG;
  if (dropout != nullptr && dropout->atom_offset < 0) return LIGDIFF_ERR_ARG;
  if (!w || !g || !h || !pos || !pair_ptr || !atom_logits || !valency || !adjacency || !workspace) return LIGDIFF_ERR_ARG;
  if (n_pairs > 0 && !bond_logits) return LIGDIFF_ERR_ARG;
  {
    const int grc = check_graph(g);
    if (grc) return grc;
  }
  const int64_t N64 = g->n_atoms;
  if (N64 <= 0 || g->n_mols <= 0 || n_pairs < 0 || n_max < g->max_mol_atoms) return LIGDIFF_ERR_ARG;
  if (N64 > 0x7fffffff / 512 || n_pairs > 0x7fffffff / 256) return LIGDIFF_ERR_ARG;   // int row indices in the GEMMs
  if (workspace_bytes < ligdiff_consistency_workspace_bytes(N64, n_pairs)) return LIGDIFF_ERR_WORKSPACE;
  const int N = (int)N64, B = g->n_mols;
  hipStream_t s = (hipStream_t)stream;
  ConsLayout L;
  cons_layout(&L, N, n_pairs);
  char* ws = (char*)workspace;
  auto F = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
  int32_t* pred = reinterpret_cast<int32_t*>(ws + L.pred);

  // --- atom types (model1.py:458-459): 256 -> 256 GELU LN -> 128 GELU -> 119
  {
    GemmCore gc = core_blocks(N, 256, 256, w[W_CONS(CONS_ATOM, CONS_W0)], 256, 1, 256);
    EpiLN<ACT_GELU, ACT_NONE, false> e;
    memset(&e, 0, sizeof e);
    e.M = N; e.eps = 1e-5f; e.C = F(L.y1); e.ldc = 256;
    e.bias[0] = w[W_CONS(CONS_ATOM, CONS_B0)]; e.gamma[0] = w[W_CONS(CONS_ATOM, CONS_G)]; e.beta[0] = w[W_CONS(CONS_ATOM, CONS_BE)];
    launch_gemm_plain<8>(gc, h, 256, e, 1, s);
    cons_trunk_tail(w, CONS_ATOM, F(L.y1), N, F(L.y2), s);
    if (d_atom.thr != 0u)
      hipLaunchKernelGGL(k_dropout_rows128, dim3(ceil_div((long)N * 128, 256)), dim3(256), 0, s, F(L.y2), (int64_t)N, d_atom);
    GemmCore g6 = core_blocks(N, 128, LD_A, w[W_CONS(CONS_ATOM, CONS_WL)], 128, 1, 128);
    EpiBiasAct<4, ACT_NONE, false> e6;
    memset(&e6, 0, sizeof e6);
    e6.M = N; e6.n_valid = LD_A; e6.col_block = 0; e6.bias[0] = w[W_CONS(CONS_ATOM, CONS_BL)]; e6.C = atom_logits;
    e6.ldc = LD_A; e6.alpha = 1.0f;
    launch_gemm_plain<4>(g6, F(L.y2), 128, e6, 1, s);
    hipLaunchKernelGGL(k_atom_softmax, dim3(ceil_div(N, 4)), dim3(256), 0, s, atom_logits, F(L.probs), pred, N);
  }
  // --- valency (model1.py:462-468): [h | probs] (375, zero-extended to 384) -> 256 GELU LN -> 128 GELU -> 1 softplus
  {
    hipLaunchKernelGGL(k_pad_cols, dim3(ceil_div(256L * 384, 256)), dim3(256), 0, s, w[W_CONS(CONS_VAL, CONS_W0)], 256, 375,
                       375, F(L.wv), 384);
    GemmCore gc = core_blocks(N, 384, 256, F(L.wv), 384, 1, 256);
    EpiLN<ACT_GELU, ACT_NONE, false> e;
    memset(&e, 0, sizeof e);
    e.M = N; e.eps = 1e-5f; e.C = F(L.y1); e.ldc = 256;
    e.bias[0] = w[W_CONS(CONS_VAL, CONS_B0)]; e.gamma[0] = w[W_CONS(CONS_VAL, CONS_G)]; e.beta[0] = w[W_CONS(CONS_VAL, CONS_BE)];
    launch_gemm_cs(gc, APtrConcat{h, F(L.probs), 256, 128, 256}, e, 1, s);
    cons_trunk_tail(w, CONS_VAL, F(L.y1), N, F(L.y2), s);
    hipLaunchKernelGGL(k_valency_out, dim3(ceil_div(N, 4)), dim3(256), 0, s, F(L.y2), w[W_CONS(CONS_VAL, CONS_WL)],
                       w[W_CONS(CONS_VAL, CONS_BL)], F(L.val), N);
    hipLaunchKernelGGL(k_valency_constrain, dim3(ceil_div((long)N * N, 256)), dim3(256), 0, s, F(L.val), pred,
                       w[W_CONS_MAXVAL], valency, (long)N);
  }
  // --- bonds over all ordered pairs (model1.py:489-551)
  {
    const long total = (long)B * n_max * n_max * 5;
    hipLaunchKernelGGL(k_adj_init, dim3(ceil_div(total, 256)), dim3(256), 0, s, adjacency, g->mol_ptr, n_max, total);
    if (n_pairs > 0) {
      const int NP = (int)n_pairs;
      const float* W0 = w[W_CONS(CONS_BOND, CONS_W0)];
      GemmCore gp;
      memset(&gp, 0, sizeof gp);
      gp.M = N; gp.K = 256; gp.n_valid = 256; gp.ldw = 516; gp.W[0] = W0; gp.W[1] = W0 + 256;
      EpiBiasAct<8, ACT_NONE, false> ep;
      memset(&ep, 0, sizeof ep);
      ep.M = N; ep.n_valid = 256; ep.col_block = 256; ep.C = F(L.P); ep.ldc = 512; ep.alpha = 1.0f;
      launch_gemm_plain<8>(gp, h, 256, ep, 2, s);
      hipLaunchKernelGGL(k_bond_pre, dim3(ceil_div(n_pairs, 4)), dim3(256), 0, s, F(L.P), pos, pred, w[W_CONS_RADII], W0,
                         w[W_CONS(CONS_BOND, CONS_B0)], w[W_CONS(CONS_BOND, CONS_G)], w[W_CONS(CONS_BOND, CONS_BE)],
                         g->mol_ptr, pair_ptr, B, n_pairs, F(L.yb1));
      cons_trunk_tail(w, CONS_BOND, F(L.yb1), NP, F(L.yb2), s);
      if (d_bond.thr != 0u)
        hipLaunchKernelGGL(k_dropout_rows128, dim3(ceil_div((long)NP * 128, 256)), dim3(256), 0, s, F(L.yb2), (int64_t)NP, d_bond);
      hipLaunchKernelGGL(k_bond_out, dim3(ceil_div(n_pairs, 4)), dim3(256), 0, s, F(L.yb2), w[W_CONS(CONS_BOND, CONS_WL)],
                         w[W_CONS(CONS_BOND, CONS_BL)], g->mol_ptr, pair_ptr, B, n_pairs, n_max, bond_logits, adjacency);
    }
  }
  LD_CHECK_LAUNCH();
  return LIGDIFF_OK;
}

extern "C" int ligdiff_consistency(const float* const* w, const ligdiff_graph* g, const float* h, const float* pos,
                                   const int64_t* pair_ptr, int64_t n_pairs, int32_t n_max, float* atom_logits,
                                   float* bond_logits, float* valency, float* adjacency, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  return ligdiff_consistency_train(w, g, h, pos, pair_ptr, n_pairs, n_max, atom_logits, bond_logits, valency, adjacency,
                                   workspace, workspace_bytes, stream, nullptr);
}
