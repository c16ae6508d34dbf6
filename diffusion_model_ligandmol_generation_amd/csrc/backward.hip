// Score-network backward: reverse sweep over the activations kept by a forward run with
// LIGDIFF_FWD_SAVE_FOR_BACKWARD.  Produces the gradient of every parameter the hot path
// reads (the same set autograd reaches in the reference, train.py:195) into caller-owned,
// pre-zeroed tensors.  Input gradients (x, pos, t, properties) are not produced: they are
// data in the training step.
#include <stdio.h>
#include <string.h>

#include "forward.h"
#include "gemm_bwd.h"
#include "gemm_dw.h"
#include "launch.h"
#include "kernels_bwd.h"
#include "weights.h"

struct BwdLayout {
  size_t dxa, n256c, n256d, G;
  size_t e256a, e4;
  size_t dmod[LD_MAXL], b256a, b256b, b256c, b256d, b512, b2048, b8, bhid;
  size_t wt;   // transposed weight block of the dX GEMM in flight (<= 1536 x 256 elements, fp32 or split bf16: 6 bytes each)
  size_t wt_all, wt_all_floats;   // every weight block the input-gradient GEMMs need transposed, filled by ONE launch
  size_t pool, pool_bytes;   // single-use buffers: everything a weight-gradient launch on the side stream reads
  size_t total;
};
// Floats of all pre-transposed weight blocks (see wt_prepare): heads 3 x 256 x 256, per attention layer FFN 2 x 256 x 1024,
// out-projection 256 x 256, in-projection 768 x 256, adaLN 1536 x 256; per E(3) layer six 256 x 256 blocks and the
// 128 x 256 coordinate gate; second time-MLP layer 256 x 512.
#define WT_ALL_FLOATS(nl) (3 * 65536 + (nl) * (2 * 262144 + 65536 + 196608 + 393216) + (nl) * (6 * 65536 + 32768) + 131072)
// Bytes of the single-use pool: one buffer per use of every tensor that a weight-gradient launch reads (heads,
// per attention layer, per E(3) layer, embeddings), each rounded like `take` below.
static size_t bwd_pool_bytes(int64_t N, int64_t E, int nl) {
  const int64_t dit[] = {N * 256, N * 1024, N * 256, N * 768};
  const int64_t e3[] = {N * 256, N * 256, E * 128, E * 256, E * 256, E * 32, E * 32, E * 32, N * 512};
  const int64_t once[] = {N * 512, N * 256, N * 256};
  size_t tot = 0;
  auto add = [&](int64_t nfloats) { tot += align_up((size_t)nfloats * 4 + 16, 256); };
  for (int l = 0; l < nl; ++l) {
    for (int64_t n : dit) add(n);
    for (int64_t n : e3) add(n);
  }
  for (int64_t n : once) add(n);
  return tot;
}
static void bwd_layout(BwdLayout* L, int64_t N, int64_t E, int32_t B) {
  const int nl = call_layer_pairs();
  size_t off = 0;
  auto take = [&](size_t nfloats) { size_t o = off; off += align_up(nfloats * 4 + 16, 256); return o; };
  L->dxa = take(N * 256); L->n256c = take(N * 256); L->n256d = take(N * 256); L->G = take(N * 3);
  L->e256a = take(E * 256); L->e4 = take(E * 4);
  for (int l = 0; l < nl; ++l) L->dmod[l] = take((size_t)B * 1536);
  L->b256a = take((size_t)B * 256); L->b256b = take((size_t)B * 256); L->b256c = take((size_t)B * 256);
  L->b256d = take((size_t)B * 256); L->b512 = take((size_t)B * 512); L->b2048 = take((size_t)B * 2048);
  L->b8 = take((size_t)B * 8); L->bhid = take((size_t)B * 512);
  L->wt = take(1536 * 256 * 3 / 2);
  L->wt_all_floats = WT_ALL_FLOATS(nl);
  L->wt_all = take(L->wt_all_floats);
  L->pool = off;
  L->pool_bytes = bwd_pool_bytes(N, E, nl);
  off += L->pool_bytes;
  L->total = off;
}

extern "C" size_t ligdiff_backward_workspace_bytes(int64_t n_atoms, int64_t n_edges, int32_t n_mols) {
  BwdLayout L;
  bwd_layout(&L, n_atoms, n_edges, n_mols);
  return L.total;
}

// dst[c, r] = src[r, c] for a [rows, cols] block with leading dimension ld (32 x 32 tiles through LDS)
__global__ __launch_bounds__(256) void k_transpose_block(const float* __restrict__ src, int rows, int cols, int ld,
                                                         float* __restrict__ dst) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int i = ty; i < 32; i += 8)
    if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = src[(size_t)(r0 + i) * ld + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < cols && r0 + tx < rows) dst[(size_t)(c0 + i) * rows + r0 + tx] = tile[tx][i];
}

// Weight-gradient launches (dW = dY^T X, bias sums, the narrow direct forms) are leaves of the backward sweep: nothing
// in it reads their results.  They run on a second stream of the calling thread, so that they fill the compute units
// the input-gradient chain leaves idle (its GEMMs are single rounds of ceil(rows / 128) workgroups: 147 of 256
// compute units at 18 756 atoms).  Ordering: a side launch is preceded by fork() (the side stream waits for
// everything enqueued on the caller's stream so far, i.e. for the producer of its dY operand); every tensor a side
// launch reads lives in a single-use buffer of the workspace pool or in the forward workspace, so the main chain
// never overwrites it; join() at the end makes the caller's stream wait for the side stream.
// LIGDIFF_DW_SIDE=0 keeps everything on the caller's stream (A/B switch).
static thread_local bool g_bwd_one_stream = false;   // set by ligdiff_backward_timed: the probed launch runs alone
struct SideStream {
  hipStream_t s = nullptr, s2 = nullptr;
  hipEvent_t* ev = nullptr;
  int next = 0;
  bool ok = true;
  static constexpr int N_EV = 32;
  struct PerDevice { hipStream_t s2 = nullptr; hipEvent_t ev[N_EV + 1] = {}; bool ready = false; };
  void init(hipStream_t main_stream) {
    s = s2 = main_stream;
    static const bool enabled = [] { const char* e = getenv("LIGDIFF_DW_SIDE"); return !(e != nullptr && e[0] == '0'); }();
    if (!enabled || g_bwd_one_stream) return;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (main_stream != nullptr && (hipStreamIsCapturing(main_stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone))
      return;   // under capture: one stream
    static thread_local PerDevice dev_ctx[64];
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
    PerDevice& d = dev_ctx[dev];
    if (!d.ready) {   // created once per thread and device, kept until the process ends
      if (hipStreamCreateWithFlags(&d.s2, hipStreamNonBlocking) != hipSuccess) { d.s2 = nullptr; return; }
      for (int i = 0; i <= N_EV; ++i)
        if (hipEventCreateWithFlags(&d.ev[i], hipEventDisableTiming) != hipSuccess) { d.s2 = nullptr; return; }
      d.ready = true;
    }
    if (d.s2 == nullptr) return;
    s2 = d.s2; ev = d.ev;
  }
  bool split() const { return s2 != s; }
  // the side stream waits for what the caller's stream holds so far; returns the stream for the leaf launches
  hipStream_t fork() {
    if (!split()) return s;
    hipEvent_t e = ev[next];
    next = (next + 1) % N_EV;
    if (hipEventRecord(e, s) != hipSuccess || hipStreamWaitEvent(s2, e, 0) != hipSuccess) ok = false;
    return s2;
  }
  void join() {
    if (!split()) return;
    if (hipEventRecord(ev[N_EV], s2) != hipSuccess || hipStreamWaitEvent(s, ev[N_EV], 0) != hipSuccess) ok = false;
  }
};

// Scratch for the transposed weight block (set per ligdiff_backward call from its workspace).
static thread_local float* g_wt_scratch = nullptr;

// All weight blocks that gemm_dx needs transposed (fp32 pipe), transposed by ONE launch at the start of the sweep
// instead of one small launch in front of every input-gradient GEMM (39 per step).  gemm_dx looks its block up by
// (pointer, leading dimension, K, columns); a block that is not in the table is transposed on the spot as before.
struct WtEntry { const float* W; int ldw, K, ncols, tile0; float* dst; };
struct WtBatch { WtEntry e[48]; int n, tiles; };
static thread_local WtBatch g_wt_table;
__global__ __launch_bounds__(256) void k_transpose_many(WtBatch b) {
  __shared__ float tile[32][33];
  int i = 0;
  while (i + 1 < b.n && (int)blockIdx.x >= b.e[i + 1].tile0) ++i;
  const WtEntry en = b.e[i];
  const int t = blockIdx.x - en.tile0, tc = en.ncols >> 5;
  const int r0 = (t / tc) * 32, c0 = (t % tc) * 32;   // rows = K index of W, columns = ncols index
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) tile[k][tx] = en.W[(size_t)(r0 + k) * en.ldw + c0 + tx];
  __syncthreads();
  for (int k = ty; k < 32; k += 8) en.dst[(size_t)(c0 + k) * en.K + r0 + tx] = tile[tx][k];
}
static const float* wt_lookup(const float* W, int ldw, int K, int ncols) {
  const WtBatch& b = g_wt_table;
  for (int i = 0; i < b.n; ++i)
    if (b.e[i].W == W && b.e[i].ldw == ldw && b.e[i].K == K && b.e[i].ncols == ncols) return b.e[i].dst;
  return nullptr;
}
static bool wt_add(const float* W, int ldw, int K, int ncols, float* base, size_t cap_floats, size_t* used) {
  WtBatch& b = g_wt_table;
  if (b.n >= 48 || (K & 31) || (ncols & 31) || *used + (size_t)K * ncols > cap_floats) return false;
  b.e[b.n] = WtEntry{W, ldw, K, ncols, b.tiles, base + *used};
  b.tiles += (K >> 5) * (ncols >> 5);
  *used += (size_t)K * ncols;
  ++b.n;
  return true;
}

// Measurement hook (ligdiff_backward_timed): the `want`-th launch of the wide weight-gradient kernel
// gemm_dw_kernel<4> of a backward pass, counted in launch order, is bracketed by two events.
struct BwdProbe { int want, seen, hits; int64_t nout, kin, rows; hipEvent_t e0, e1; };
static thread_local BwdProbe* g_bwd_probe = nullptr;

// dW[nout, kin] += dY[rows, nout]^T X[rows, kin] (+ db) for narrow / unaligned shapes (k_dw_direct)
static void gemm_dw_direct(const float* dY, int lddy, int nout, const float* X, int ldx, int kin, int rows, float* dW,
                           int lddw, hipStream_t s, float* db = nullptr) {
  if (rows <= 0) return;
  const int tiles = ceil_div(nout, 32) * ceil_div(kin, 32);
  int slices = ceil_div(2048, tiles);                   // ~2048 waves in flight
  int rps = (int)align_up(ceil_div(rows, slices), 32);
  if (rps < 128) rps = 128;
  slices = ceil_div(rows, rps);
  hipLaunchKernelGGL(k_dw_direct, dim3(ceil_div(tiles * slices, 4)), dim3(256), 0, s, dY, lddy, nout, X, ldx, kin, rows, rps,
                     dW, lddw, db);
}

// C[M, ncols] (+)= dY[M, K] * W[K, ncols]  [* act'(pre)]     (W is the forward weight [out=K, in])
// When the shape allows (ncols a multiple of 256, K of 16, aligned rows) the weight block is
// transposed into scratch - a few hundred KB - and the product runs on the forward pass's LDS-DMA
// kernel, whose operands both need the reduction index contiguous.
// drop (column-split path only, the caller checks): dropout site whose mask multiplies the result.
template <int NT, int MUL, bool ACCUM>
static void gemm_dx(const float* dY, int lddy, int M, int K, const float* W, int ldw, int ncols, float* C, int ldc,
                    const float* pre, int ldp, hipStream_t s, const DropSite* drop = nullptr) {
  if (M <= 0) return;
  DropSite ds;
  memset(&ds, 0, sizeof ds);
  if (drop != nullptr) ds = *drop;
  if (NT == 8 && use_cs() && g_wt_scratch != nullptr && ncols % 256 == 0 && K % G2_BK == 0 && lddy % 4 == 0 &&
      ncols / 256 <= 8 && (size_t)K * ncols <= 1536 * 256) {
    if (split_gemm_enabled()) {   // split-operand mode: W^T goes straight into the split-bf16 block layout
      uint16_t* ws = reinterpret_cast<uint16_t*>(g_wt_scratch);
      hipLaunchKernelGGL(k_split_weight_t, dim3(ceil_div((K / 16) * ncols, 256)), dim3(256), 0, s, W, ldw, K, ncols, ws);
      GemmCore g;
      memset(&g, 0, sizeof g);
      g.M = M; g.K = K; g.n_valid = 256; g.ldw = K;
      SplitW sw;
      memset(&sw, 0, sizeof sw);
      for (int b = 0; b < ncols / 256; ++b) sw.blk[b] = ws + (size_t)b * (K / 16) * G4_CHUNK_ELEMS(4);
      EpiBwdStore<8, MUL, ACCUM> e{M, ncols, C, ldc, pre, ldp, 1.0f, ds};
      hipLaunchKernelGGL((gemm_cs_split_kernel<APtrPlain, EpiBwdStore<8, MUL, ACCUM>, 4>), dim3(ceil_div(M, 128), ncols / 256),
                         dim3(512), 0, s, g, sw, APtrPlain{dY, lddy}, e);
      return;
    }
    const float* wt = wt_lookup(W, ldw, K, ncols);
    if (wt == nullptr) {
      hipLaunchKernelGGL(k_transpose_block, dim3(ceil_div(ncols, 32), ceil_div(K, 32)), dim3(256), 0, s, W, K, ncols, ldw,
                         g_wt_scratch);
      wt = g_wt_scratch;
    }
    GemmCore g = core_blocks(M, K, 256, wt, K, ncols / 256, 256);
    EpiBwdStore<8, MUL, ACCUM> e{M, ncols, C, ldc, pre, ldp, 1.0f, ds};
    launch_gemm_cs(g, APtrPlain{dY, lddy}, e, ncols / 256, s);
    return;
  }
  OpRow a{dY, lddy, M, K};
  OpK b{W, ldw, ncols, K};
  EpiBwdStore<NT, MUL, ACCUM> e{M, ncols, C, ldc, pre, ldp, 1.0f, ds};   // (this path ignores ds: see check_dropout)
  dim3 grid(ceil_div(M, G_BM), ceil_div(ncols, NT * 32), 1);
  hipLaunchKernelGGL((gemm_bwd_kernel<NT, OpRow, OpK, EpiBwdStore<NT, MUL, ACCUM>>), grid, dim3(256), 0, s, a, b,
                     (int)align_up(K, 32), e);
}

// dW[nout, kin] += dY[rows, nout]^T * X[rows, kin]   (split over rows, float atomics)
// db (nullable, LDS-DMA path only - see the return value): bias gradient sum_rows dY, from the same pass.
// Returns true when db was produced here (the caller then skips its separate column-sum launch).
template <int NT>
static bool gemm_dw(const float* dY, int lddy, int nout, const float* X, int ldx, int kin, int rows, float* dW,
                    int lddw, hipStream_t s, float* db = nullptr) {
  if (rows <= 0) return false;
  if (use_cs() && kin % 256 == 0 && lddy % 4 == 0 && ldx % 4 == 0 && nout % 4 == 0 && nout >= 4) {
    // small gradients use 128-column tiles: twice the tiles, so half the slices per tile
    const bool narrow = (long)ceil_div(nout, 128) * (kin / 256) <= 4;
    const int bx = ceil_div(nout, 128), by = narrow ? kin / 128 : kin / 256;
    // every slice of the row range adds a full tile with atomics, so the slice count is kept
    // near 1.5 workgroups per CU (measured best of 128 / 256 / 384 / 512 on the training step)
    static const int target_blocks = getenv("LIGDIFF_DW_BLOCKS") ? atoi(getenv("LIGDIFF_DW_BLOCKS")) : 384;
    int splits = ceil_div(target_blocks, bx * by);
    const int max_splits = ceil_div(rows, 64);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    const int m_split = (int)align_up(ceil_div(rows, splits), 16);
    splits = ceil_div(rows, m_split);
    GemmDwArgs a{dY, lddy, nout, X, ldx, kin, rows, m_split, dW, lddw, db};
    if (narrow) {
      hipLaunchKernelGGL(gemm_dw_kernel<2>, dim3(bx, by, splits), dim3(512), 0, s, a);
    } else {
      BwdProbe* pr = g_bwd_probe;
      const bool hit = pr != nullptr && pr->seen++ == pr->want;
      if (hit) (void)hipEventRecord(pr->e0, s);
      hipLaunchKernelGGL(gemm_dw_kernel<4>, dim3(bx, by, splits), dim3(512), 0, s, a);
      if (hit) {
        (void)hipEventRecord(pr->e1, s);
        pr->hits++; pr->nout = nout; pr->kin = kin; pr->rows = rows;
      }
    }
    return db != nullptr;
  }
  OpK a{dY, lddy, nout, rows};
  OpK b{X, ldx, kin, rows};
  EpiBwdAtomic<NT> e{nout, kin, dW, lddw};
  const int bx = ceil_div(nout, G_BM), by = ceil_div(kin, NT * 32);
  int splits = ceil_div(1024, bx * by);
  const int max_splits = ceil_div(rows, 128);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  const int k_split = (int)align_up(ceil_div(rows, splits), 32);
  splits = ceil_div(rows, k_split);
  dim3 grid(bx, by, splits);
  hipLaunchKernelGGL((gemm_bwd_kernel<NT, OpK, OpK, EpiBwdAtomic<NT>>), grid, dim3(256), 0, s, a, b, k_split, e);
  return false;
}

static void colsum(const float* Y, int ld, int M, int ncols, const float* rowscale, float* out, hipStream_t s) {
  if (M <= 0) return;
  hipLaunchKernelGGL(k_colsum, dim3(ceil_div(M, 128), ceil_div(ncols, 256)), dim3(256), 0, s, Y, ld, M, ncols, rowscale, out);
}

template <int PRE, int POST>
static void ln_bwd(const float* u, int ldu, const float* gamma, const float* beta, float eps, const float* dy, int ldy,
                   const float* dy2, int ld2, const int32_t* idx, float* du, int ldo, float* dgamma, float* dbeta,
                   int M, hipStream_t s) {
  if (M <= 0) return;
  const int rpw = M > 16384 ? 16 : 8;   // measured on edge-row calls: 8 / 16 / 32 / 64 rows per wave -> 95 / 75 / 95 / 150 us
  hipLaunchKernelGGL((k_ln_bwd<PRE, POST>), dim3(ceil_div(M, 4 * rpw)), dim3(256), 0, s, u, ldu, gamma, beta, eps, dy, ldy, dy2,
                     ld2, idx, du, ldo, dgamma, dbeta, M, rpw);
}

__global__ void k_gelu_elem(const float* __restrict__ in, float* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = gelu_exact(in[i]);
}

extern "C" int ligdiff_backward_train(const float* const* w, float* const* gr, const ligdiff_graph* g, const float* x,
                                      const float* pos, const float* properties, int flags, const float* d_atom,
                                      const float* d_pos, const float* d_prop, const void* fwd_workspace,
                                      void* workspace, size_t workspace_bytes, void* stream,
                                      const ligdiff_dropout* dropout) {
  if (!w || !gr || !g || !x || !pos || !d_atom || !d_pos || !fwd_workspace || !workspace) return LIGDIFF_ERR_ARG;
  {
    const int drc = check_dropout(dropout);
    if (drc) return drc;
  }
  {
    const int grc = check_graph(g);
    if (grc) return grc;
  }
  if (flags & ~FWD_PUBLIC_FLAGS) return LIGDIFF_ERR_ARG;
  if (!(flags & LIGDIFF_FWD_SAVE_FOR_BACKWARD) || (flags & LIGDIFF_FWD_T_BROADCAST)) return LIGDIFF_ERR_ARG;
  // every slot of the score network needs a gradient buffer; the frequency table and the consistency
  // module (slots from W_CONS_0 on: no loss reaches it) have none
  const int NL = call_layer_pairs();   // E(3) layers = attention layers of this model
  for (int i = 0; i < W_CONS_0; ++i)
    if (weight_slot_layer(i) < NL && (w[i] == nullptr || (i != W_TIME_FREQS && gr[i] == nullptr))) return LIGDIFF_ERR_ARG;
  const int N = (int)g->n_atoms, E = (int)g->n_edges, B = g->n_mols;
  FwdLayout F;
  fwd_layout(&F, N, E, B, flags);
  BwdLayout L;
  bwd_layout(&L, N, E, B);
  if (workspace_bytes < L.total) return LIGDIFF_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const char* fws = (const char*)fwd_workspace;
  char* ws = (char*)workspace;
  auto A = [&](size_t off) { return reinterpret_cast<const float*>(fws + off); };  // saved activation
  auto T = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };         // backward scratch
  const int nb4N = ceil_div(N, 4), nb4B = ceil_div(B, 4);
  float* dxa = T(L.dxa);
  g_wt_scratch = T(L.wt);
  g_wt_table.n = 0; g_wt_table.tiles = 0;
  struct WtTableScope { ~WtTableScope() { g_wt_table.n = 0; } } wt_table_scope;   // entries point into this call's workspace
  static const bool wt_batch = [] { const char* e = getenv("LIGDIFF_WT_BATCH"); return !(e != nullptr && e[0] == '0'); }();   // A/B switch
  if (wt_batch && use_cs() && !split_gemm_enabled()) {
    float* base = T(L.wt_all);
    size_t used = 0;
    auto add = [&](const float* W, int ldw, int K, int ncols) { (void)wt_add(W, ldw, K, ncols, base, L.wt_all_floats, &used); };
    for (int h = 0; h < 3; ++h) add(w[W_HEAD(h, HEAD_W0)], 256, 256, 256);
    for (int l = 0; l < NL; ++l) {
      add(w[W_DIT(l, DIT_FF3_W)], 1024, 256, 1024);
      add(w[W_DIT(l, DIT_FF0_W)], 256, 1024, 256);
      add(w[W_DIT(l, DIT_OUT_W)], 256, 256, 256);
      add(w[W_DIT(l, DIT_INP_W)], 256, 768, 256);
      add(w[W_DIT(l, DIT_ADA_W)], 256, 1536, 256);
      add(w[W_E3(l, E3_NN2_W)], 256, 256, 256);
      add(w[W_E3(l, E3_NN0_W)], 512, 256, 256);
      add(w[W_E3(l, E3_NN0_W)] + 256, 512, 256, 256);
      add(w[W_E3(l, E3_CN0_W)], 256, 128, 256);
      add(w[W_E3(l, E3_EN2_W)], 256, 256, 256);
      add(w[W_E3(l, E3_EN0_W)], 544, 256, 256);
      add(w[W_E3(l, E3_EN0_W)] + 256, 544, 256, 256);
    }
    add(w[W_T2_W], 512, 256, 512);
    if (g_wt_table.n > 0) hipLaunchKernelGGL(k_transpose_many, dim3(g_wt_table.tiles), dim3(256), 0, s, g_wt_table);
  }
  SideStream side;
  side.init(s);
  struct JoinAtExit {   // also on the error returns below: nothing stays behind on the side stream unobserved
    SideStream& st;
    ~JoinAtExit() { st.join(); }
  } join_at_exit{side};
  size_t pool_used = 0;
  bool pool_ok = true;
  // a buffer nobody writes again in this call once its producer ran (see SideStream)
  auto fresh = [&](int64_t nfloats) {
    const size_t need = align_up((size_t)nfloats * 4 + 16, 256);
    if (pool_used + need > L.pool_bytes) { pool_ok = false; return T(L.pool); }
    float* p = T(L.pool + pool_used);
    pool_used += need;
    return p;
  };

  // ------------------------------------------------------------------ heads (model1.py:741-746)
  const float* x_last = A(F.x2[NL - 1]);
  float* dyh = fresh((int64_t)N * 512);
  gemm_dx<8, MUL_NONE, false>(d_atom, LD_A, N, LD_A, w[W_HEAD(0, HEAD_W3)], 256, 256, dyh, 512, nullptr, 0, s);
  hipLaunchKernelGGL(k_small_dx<3>, dim3(nb4N), dim3(256), 0, s, d_pos, w[W_HEAD(1, HEAD_W3)], dyh, 512, 256, N);
  {
    hipStream_t sd = side.fork();   // (reads the caller's d_atom / d_pos and the forward workspace only)
    gemm_dw_direct(d_atom, LD_A, LD_A, A(F.yh), 512, 256, N, gr[W_HEAD(0, HEAD_W3)], 256, sd, gr[W_HEAD(0, HEAD_B3)]);
    hipLaunchKernelGGL(k_small_dw<3>, dim3(ceil_div(N, 128)), dim3(256), 0, sd, d_pos, A(F.yh), 512, 256,
                       gr[W_HEAD(1, HEAD_W3)], gr[W_HEAD(1, HEAD_B3)], N);
  }
  for (int h = 0; h < 2; ++h)
    ln_bwd<1, 0>(A(F.u_head) + h * 256, 512, w[W_HEAD(h, HEAD_G)], w[W_HEAD(h, HEAD_BE)], 1e-5f, dyh + h * 256, 512,
                 nullptr, 0, nullptr, dyh + h * 256, 512, gr[W_HEAD(h, HEAD_G)], gr[W_HEAD(h, HEAD_BE)], N, s);
  {
    hipStream_t sd = side.fork();
    for (int h = 0; h < 2; ++h)
      if (!gemm_dw<8>(dyh + h * 256, 512, 256, x_last, 256, 256, N, gr[W_HEAD(h, HEAD_W0)], 256, sd, gr[W_HEAD(h, HEAD_B0)]))
        colsum(dyh + h * 256, 512, N, 256, nullptr, gr[W_HEAD(h, HEAD_B0)], sd);
  }
  gemm_dx<8, MUL_NONE, false>(dyh, 512, N, 256, w[W_HEAD(0, HEAD_W0)], 256, 256, dxa, 256, nullptr, 0, s);
  gemm_dx<8, MUL_NONE, true>(dyh + 256, 512, N, 256, w[W_HEAD(1, HEAD_W0)], 256, 256, dxa, 256, nullptr, 0, s);
  if (d_prop != nullptr && !(flags & LIGDIFF_FWD_SKIP_PROP_HEAD)) {
    float* dph = T(L.b256a);
    hipLaunchKernelGGL(k_small_dx<15>, dim3(nb4B), dim3(256), 0, s, d_prop, w[W_HEAD(2, HEAD_W3)], dph, 256, 0, B);
    hipLaunchKernelGGL(k_small_dw<15>, dim3(ceil_div(B, 128)), dim3(256), 0, s, d_prop, A(F.ph), 256, 0,
                       gr[W_HEAD(2, HEAD_W3)], gr[W_HEAD(2, HEAD_B3)], B);
    ln_bwd<1, 0>(A(F.u_ph), 256, w[W_HEAD(2, HEAD_G)], w[W_HEAD(2, HEAD_BE)], 1e-5f, dph, 256, nullptr, 0, nullptr, dph,
                 256, gr[W_HEAD(2, HEAD_G)], gr[W_HEAD(2, HEAD_BE)], B, s);
    if (!gemm_dw<8>(dph, 256, 256, A(F.pooled), 256, 256, B, gr[W_HEAD(2, HEAD_W0)], 256, s, gr[W_HEAD(2, HEAD_B0)]))
      colsum(dph, 256, B, 256, nullptr, gr[W_HEAD(2, HEAD_B0)], s);
    float* dpool = T(L.b256b);
    gemm_dx<8, MUL_NONE, false>(dph, 256, B, 256, w[W_HEAD(2, HEAD_W0)], 256, 256, dpool, 256, nullptr, 0, s);
    hipLaunchKernelGGL(k_pool_bwd, dim3(nb4N), dim3(256), 0, s, dpool, g->mol_ptr, g->atom_mol, dxa, N);
  }
  float* G = T(L.G);  // gradient w.r.t. the accumulated coordinate update (model1.py:742)
  if (hipMemcpyAsync(G, d_pos, (size_t)N * 3 * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return LIGDIFF_ERR_LAUNCH;

  // ------------------------------------------------------------------ DiT layers (model1.py:279-302)
  const size_t att_lds_hpb4 = (size_t)4 * (g->max_mol_atoms <= 64 ? 64 : 128) * 130 * 4;
  for (int l = NL - 1; l >= 0; --l) {
    const float* x_in = (l == 0) ? A(F.h_e3[NL - 1]) : A(F.x2[l - 1]);
    const float* mod = A(F.mod[l]);
    float* dmod = T(L.dmod[l]);
    float* t256b = fresh((int64_t)N * 256);
    float* t256c = T(L.n256c);
    DropSite d_att, d_hid, d_out;
    if (!make_drop_site(dropout, l, LIGDIFF_DROP_ATTENTION, &d_att) || !make_drop_site(dropout, l, LIGDIFF_DROP_FFN_HIDDEN, &d_hid) ||
        !make_drop_site(dropout, l, LIGDIFF_DROP_FFN_OUT, &d_out))
      return LIGDIFF_ERR_ARG;
    // x2 = x1 + gate_mlp * Dropout(ffh W3^T + b)   (fproj holds the masked projection)
    hipLaunchKernelGGL(k_gate_bwd, dim3(B), dim3(256), 0, s, dxa, A(F.fproj[l]), mod, 1536, 1280, g->mol_ptr, B, t256b, dmod,
                       d_out);
    {
      hipStream_t sd = side.fork();
      if (!gemm_dw<8>(t256b, 256, 256, A(F.ffh[l]), 1024, 1024, N, gr[W_DIT(l, DIT_FF3_W)], 1024, sd, gr[W_DIT(l, DIT_FF3_B)]))
        colsum(t256b, 256, N, 256, nullptr, gr[W_DIT(l, DIT_FF3_B)], sd);
    }
    float* dffh = fresh((int64_t)N * 1024);
    gemm_dx<8, MUL_GELU_GRAD, false>(t256b, 256, N, 256, w[W_DIT(l, DIT_FF3_W)], 1024, 1024, dffh, 1024,
                                     A(F.ffh_pre[l]), 1024, s, &d_hid);   // ffh = Dropout(GELU(pre))
    {
      hipStream_t sd = side.fork();
      if (!gemm_dw<8>(dffh, 1024, 1024, A(F.xn2[l]), 256, 256, N, gr[W_DIT(l, DIT_FF0_W)], 256, sd, gr[W_DIT(l, DIT_FF0_B)]))
        colsum(dffh, 1024, N, 1024, nullptr, gr[W_DIT(l, DIT_FF0_B)], sd);
    }
    gemm_dx<8, MUL_NONE, false>(dffh, 1024, N, 1024, w[W_DIT(l, DIT_FF0_W)], 256, 256, t256c, 256, nullptr, 0, s);
    hipLaunchKernelGGL(k_lnmod_bwd, dim3(B), dim3(256), 0, s, A(F.x1[l]), t256c, dxa, mod, 1536, 768, 1024,
                       g->mol_ptr, B, dxa, dmod);
    // x1 = x_in + gate_msa * (att Wo^T + b)
    t256b = fresh((int64_t)N * 256);
    hipLaunchKernelGGL(k_gate_bwd, dim3(B), dim3(256), 0, s, dxa, A(F.oproj[l]), mod, 1536, 512, g->mol_ptr, B, t256b, dmod,
                       DropSite{});
    {
      hipStream_t sd = side.fork();
      if (!gemm_dw<8>(t256b, 256, 256, A(F.att[l]), 256, 256, N, gr[W_DIT(l, DIT_OUT_W)], 256, sd, gr[W_DIT(l, DIT_OUT_B)]))
        colsum(t256b, 256, N, 256, nullptr, gr[W_DIT(l, DIT_OUT_B)], sd);
    }
    gemm_dx<8, MUL_NONE, false>(t256b, 256, N, 256, w[W_DIT(l, DIT_OUT_W)], 256, 256, t256c, 256, nullptr, 0, s);
    float* dqkv = fresh((int64_t)N * 768);
    if (d_att.thr != 0u) {   // probabilities were dropout-masked in the forward pass
      const int tk = (g->max_mol_atoms + 31) / 32;
      const float* qkv_l = A(F.qkv[l]);
      const float* att_l = A(F.att[l]);
      const float* lse_l = A(F.lse[l]);
      if (tk == 1)
        hipLaunchKernelGGL((k_attention_bwd_mfma<1, 4, true>), dim3(ceil_div((long)B * 8, 4)), dim3(512), 4 * 32 * 134 * 4, s,
                           qkv_l, att_l, t256c, lse_l, g->mol_ptr, dqkv, B, d_att);
      else if (tk == 2)
        hipLaunchKernelGGL((k_attention_bwd_mfma<2, 4, true>), dim3(ceil_div((long)B * 8, 4)), dim3(512), 4 * 64 * 134 * 4, s,
                           qkv_l, att_l, t256c, lse_l, g->mol_ptr, dqkv, B, d_att);
      else if (tk == 3)
        hipLaunchKernelGGL((k_attention_bwd_mfma<3, 2, true>), dim3(ceil_div((long)B * 8, 2)), dim3(256), 2 * 96 * 134 * 4, s,
                           qkv_l, att_l, t256c, lse_l, g->mol_ptr, dqkv, B, d_att);
      else
        hipLaunchKernelGGL((k_attention_bwd_mfma<4, 2, true>), dim3(ceil_div((long)B * 8, 2)), dim3(256), 2 * 128 * 134 * 4, s,
                           qkv_l, att_l, t256c, lse_l, g->mol_ptr, dqkv, B, d_att);
    } else if (getenv("LIGDIFF_ATT_BWD_V1") != nullptr) {   // scalar reference kernel (A/B switch used by the tests)
      if (g->max_mol_atoms <= 64)
        hipLaunchKernelGGL(k_attention_bwd, dim3(B * 2), dim3(256), att_lds_hpb4, s, A(F.qkv[l]), A(F.att[l]), t256c,
                           A(F.lse[l]), g->mol_ptr, dqkv, 64);
      else
        hipLaunchKernelGGL(k_attention_bwd, dim3(B * 4), dim3(128), att_lds_hpb4 / 2, s, A(F.qkv[l]), A(F.att[l]), t256c,
                           A(F.lse[l]), g->mol_ptr, dqkv, 128);
    } else {
      const int tk = (g->max_mol_atoms + 31) / 32;
      const float* qkv_l = A(F.qkv[l]);
      const float* att_l = A(F.att[l]);
      const float* lse_l = A(F.lse[l]);
      if (tk == 1)
        hipLaunchKernelGGL((k_attention_bwd_mfma<1, 4>), dim3(ceil_div((long)B * 8, 4)), dim3(512), 4 * 32 * 134 * 4, s, qkv_l,
                           att_l, t256c, lse_l, g->mol_ptr, dqkv, B);
      else if (tk == 2)
        hipLaunchKernelGGL((k_attention_bwd_mfma<2, 4>), dim3(ceil_div((long)B * 8, 4)), dim3(512), 4 * 64 * 134 * 4, s, qkv_l,
                           att_l, t256c, lse_l, g->mol_ptr, dqkv, B);
      else if (tk == 3)
        hipLaunchKernelGGL((k_attention_bwd_mfma<3, 2>), dim3(ceil_div((long)B * 8, 2)), dim3(256), 2 * 96 * 134 * 4, s, qkv_l,
                           att_l, t256c, lse_l, g->mol_ptr, dqkv, B);
      else
        hipLaunchKernelGGL((k_attention_bwd_mfma<4, 2>), dim3(ceil_div((long)B * 8, 2)), dim3(256), 2 * 128 * 134 * 4, s, qkv_l,
                           att_l, t256c, lse_l, g->mol_ptr, dqkv, B);
    }
    {
      hipStream_t sd = side.fork();
      if (!gemm_dw<8>(dqkv, 768, 768, A(F.xn1[l]), 256, 256, N, gr[W_DIT(l, DIT_INP_W)], 256, sd, gr[W_DIT(l, DIT_INP_B)]))
        colsum(dqkv, 768, N, 768, nullptr, gr[W_DIT(l, DIT_INP_B)], sd);
    }
    gemm_dx<8, MUL_NONE, false>(dqkv, 768, N, 768, w[W_DIT(l, DIT_INP_W)], 256, 256, t256c, 256, nullptr, 0, s);
    hipLaunchKernelGGL(k_lnmod_bwd, dim3(B), dim3(256), 0, s, x_in, t256c, dxa, mod, 1536, 0, 256, g->mol_ptr, B,
                       dxa, dmod);
  }

  // ------------------------------------------------------------------ E(3) layers (model1.py:86-114, :721-725)
  for (int l = NL - 1; l >= 0; --l) {
    const float* h_in = (l == 0) ? A(F.h0) : A(F.h_e3[l - 1]);
    const float* W1 = w[W_E3(l, E3_EN0_W)];
    float* dW1 = gr[W_E3(l, E3_EN0_W)];
    float* t256b = fresh((int64_t)N * 256);
    float* t256c = fresh((int64_t)N * 256);
    float* dagg = T(L.n256d);
    // h_out = h_in + LN(hid Wn2^T + b)
    ln_bwd<0, 0>(A(F.u_h[l]), 256, w[W_E3(l, E3_NN3_G)], w[W_E3(l, E3_NN3_B)], 1e-5f, dxa, 256, nullptr, 0, nullptr,
                 t256b, 256, gr[W_E3(l, E3_NN3_G)], gr[W_E3(l, E3_NN3_B)], N, s);
    {
      hipStream_t sd = side.fork();
      if (!gemm_dw<8>(t256b, 256, 256, A(F.hid[l]), 256, 256, N, gr[W_E3(l, E3_NN2_W)], 256, sd, gr[W_E3(l, E3_NN2_B)]))
        colsum(t256b, 256, N, 256, nullptr, gr[W_E3(l, E3_NN2_B)], sd);
    }
    gemm_dx<8, MUL_SILU_GRAD, false>(t256b, 256, N, 256, w[W_E3(l, E3_NN2_W)], 256, 256, t256c, 256, A(F.hid_pre[l]),
                                     256, s);
    {
      hipStream_t sd = side.fork();
      gemm_dw<8>(t256c, 256, 256, h_in, 256, 256, N, gr[W_E3(l, E3_NN0_W)], 512, sd);
      if (!gemm_dw<8>(t256c, 256, 256, A(F.agg[l]), 256, 256, N, gr[W_E3(l, E3_NN0_W)] + 256, 512, sd, gr[W_E3(l, E3_NN0_B)]))
        colsum(t256c, 256, N, 256, nullptr, gr[W_E3(l, E3_NN0_B)], sd);
    }
    gemm_dx<8, MUL_NONE, true>(t256c, 256, N, 256, w[W_E3(l, E3_NN0_W)], 512, 256, dxa, 256, nullptr, 0, s);
    gemm_dx<8, MUL_NONE, false>(t256c, 256, N, 256, w[W_E3(l, E3_NN0_W)] + 256, 512, 256, dagg, 256, nullptr, 0, s);
    if (E > 0) {
      float* drel = T(L.e4);
      float* dhc = fresh((int64_t)E * 128);
      float* e256a = T(L.e256a);
      float* e256b = fresh((int64_t)E * 256);
      float* dpre = fresh((int64_t)E * 256);
      // coordinate gate
      hipLaunchKernelGGL(k_coord_bwd, dim3(512), dim3(256), 0, s, G, g->src, g->dst, A(F.rel[l]), A(F.cw[l]),
                         A(F.hc_pre[l]), w[W_E3(l, E3_CN2_W)], drel, dhc, gr[W_E3(l, E3_CN2_W)], gr[W_E3(l, E3_CN2_B)], E);
      {
        hipStream_t sd = side.fork();
        if (!gemm_dw<8>(dhc, 128, 128, A(F.msg[l]), 256, 256, E, gr[W_E3(l, E3_CN0_W)], 256, sd, gr[W_E3(l, E3_CN0_B)]))
          colsum(dhc, 128, E, 128, nullptr, gr[W_E3(l, E3_CN0_B)], sd);
      }
      gemm_dx<8, MUL_NONE, false>(dhc, 128, E, 128, w[W_E3(l, E3_CN0_W)], 256, 256, e256a, 256, nullptr, 0, s);
      // msg = LN(a1 W2^T + b); its gradient = coordinate branch + aggregated branch (dagg[dst])
      ln_bwd<0, 0>(A(F.u_m[l]), 256, w[W_E3(l, E3_EN3_G)], w[W_E3(l, E3_EN3_B)], 1e-5f, e256a, 256, dagg, 256, g->dst,
                   e256b, 256, gr[W_E3(l, E3_EN3_G)], gr[W_E3(l, E3_EN3_B)], E, s);
      {
        hipStream_t sd = side.fork();
        if (!gemm_dw<8>(e256b, 256, 256, A(F.a1[l]), 256, 256, E, gr[W_E3(l, E3_EN2_W)], 256, sd, gr[W_E3(l, E3_EN2_B)]))
          colsum(e256b, 256, E, 256, nullptr, gr[W_E3(l, E3_EN2_B)], sd);
      }
      gemm_dx<8, MUL_SILU_GRAD, false>(e256b, 256, E, 256, w[W_E3(l, E3_EN2_W)], 256, 256, dpre, 256, A(F.pre_e[l]),
                                       256, s);
      // pre = Pa[src] + Pb[dst] + Wc rbf + b1.  The bias gradient sum_e dpre[e] equals the column sum of
      // the per-destination gather dP[:, 256:512] (every edge has exactly one destination atom), which
      // the weight-gradient GEMM on that block produces in the same pass (N rows instead of E).
      gemm_dw_direct(dpre, 256, 256, A(F.rbf[l]), 32, 32, E, dW1 + 512, 544, side.fork());
      float* drbf = fresh((int64_t)E * 32);
      gemm_dx<4, MUL_NONE, false>(dpre, 256, E, 256, W1 + 512, 544, 32, drbf, 32, nullptr, 0, s);
      float* dr1pre = fresh((int64_t)E * 32);
      float* r1 = fresh((int64_t)E * 32);
      hipLaunchKernelGGL(k_radial_bwd, dim3(ceil_div(E, 256)), dim3(256), 0, s, drbf, A(F.r1pre[l]), A(F.dist[l]),
                         A(F.rel[l]), w[W_E3(l, E3_RB0_W)], w[W_E3(l, E3_RB2_W)], drel, dr1pre, r1, E);
      {
        hipStream_t sd = side.fork();
        gemm_dw_direct(drbf, 32, 32, r1, 32, 32, E, gr[W_E3(l, E3_RB2_W)], 32, sd);
        hipLaunchKernelGGL(k_radial_colsums, dim3(ceil_div(E, 256)), dim3(256), 0, sd, drbf, dr1pre, A(F.dist[l]),
                           gr[W_E3(l, E3_RB2_B)], gr[W_E3(l, E3_RB0_W)], gr[W_E3(l, E3_RB0_B)], E);
      }
      float* dP = fresh((int64_t)N * 512);
      hipLaunchKernelGGL(k_dp_gather, dim3(nb4N), dim3(256), 0, s, dpre, g->dst_ptr, g->src_ptr, g->src_slot, dP, N);
      {
        hipStream_t sd = side.fork();
        gemm_dw<8>(dP, 512, 256, h_in, 256, 256, N, dW1, 544, sd);
        if (!gemm_dw<8>(dP + 256, 512, 256, h_in, 256, 256, N, dW1 + 256, 544, sd, gr[W_E3(l, E3_EN0_B)]))
          colsum(dpre, 256, E, 256, nullptr, gr[W_E3(l, E3_EN0_B)], sd);
      }
      gemm_dx<8, MUL_NONE, true>(dP, 512, N, 256, W1, 544, 256, dxa, 256, nullptr, 0, s);
      gemm_dx<8, MUL_NONE, true>(dP + 256, 512, N, 256, W1 + 256, 544, 256, dxa, 256, nullptr, 0, s);
      if (l > 0)  // layer 0 reads the data coordinates: no parameter lies upstream of them
        hipLaunchKernelGGL(k_pos_scatter_bwd, dim3(nb4N), dim3(256), 0, s, drel, g->dst_ptr, g->src_ptr, g->src_slot, G, N);
    }
  }

  // ------------------------------------------------------------------ embeddings (model1.py:709-718)
  {
    float* du = fresh((int64_t)N * 256);
    ln_bwd<0, 1>(A(F.ua), 256, w[W_ATOM_G], w[W_ATOM_BE], 1e-5f, dxa, 256, nullptr, 0, nullptr, du, 256, gr[W_ATOM_G],
                 gr[W_ATOM_BE], N, s);
    gemm_dw_direct(du, 256, 256, x, LD_A, LD_A, N, gr[W_ATOM_W], LD_A, side.fork(), gr[W_ATOM_B]);
    float* up = T(L.n256c);
    float* dup = fresh((int64_t)N * 256);
    hipLaunchKernelGGL(k_pos_embed_pre, dim3(nb4N), dim3(256), 0, s, pos, w[W_POS_W], w[W_POS_B], up, N);
    ln_bwd<0, 1>(up, 256, w[W_POS_G], w[W_POS_BE], 1e-5f, dxa, 256, nullptr, 0, nullptr, dup, 256, gr[W_POS_G],
                 gr[W_POS_BE], N, s);
    {
      hipStream_t sd = side.fork();
      hipLaunchKernelGGL(k_pos_embed_dw, dim3(ceil_div(N, 128)), dim3(256), 0, sd, dup, pos, gr[W_POS_W], N);
      colsum(dup, 256, N, 256, nullptr, gr[W_POS_B], sd);
    }
  }
  float* dmolsum = T(L.b256a);  // sum over each molecule's atoms of dh0: gradient of t_emb[mol] and prop_emb[mol]
  hipLaunchKernelGGL(k_mol_sum, dim3(nb4B), dim3(256), 0, s, dxa, g->mol_ptr, B, dmolsum);

  // ------------------------------------------------------------------ PILOT conditioning (model1.py:335-359)
  if (flags & LIGDIFF_FWD_PROPERTIES) {
    if (flags & LIGDIFF_FWD_NULL_COND) {
      colsum(dmolsum, 256, B, 256, nullptr, gr[W_NULL_EMB], s);
    } else {
      if (properties == nullptr) return LIGDIFF_ERR_ARG;
      float* dstk = T(L.b2048);
      float* dz = T(L.b8);
      float* dgpre = T(L.b256d);
      float* hid = T(L.bhid);
      hipLaunchKernelGGL(k_pilot_combine_bwd, dim3(B), dim3(256), 0, s, dmolsum, A(F.pstack), A(F.pz), A(F.pw), dstk, dz);
      hipLaunchKernelGGL(k_pilot_dw2, dim3(ceil_div(B, 64)), dim3(256), 0, s, dz, A(F.pg), B, gr[W_PAR2_W], gr[W_PAR2_B]);
      hipLaunchKernelGGL(k_pilot_dg, dim3(B), dim3(256), 0, s, dz, w[W_PAR2_W], A(F.pg_pre), dgpre);
      if (!gemm_dw<8>(dgpre, 256, 256, A(F.pstack), 2048, 2048, B, gr[W_PAR0_W], 2048, s, gr[W_PAR0_B]))
        colsum(dgpre, 256, B, 256, nullptr, gr[W_PAR0_B], s);
      gemm_dx<8, MUL_NONE, true>(dgpre, 256, B, 256, w[W_PAR0_W], 2048, 2048, dstk, 2048, nullptr, 0, s);
      hipLaunchKernelGGL(k_gelu_elem, dim3(ceil_div((int64_t)B * 512, 256)), dim3(256), 0, s, A(F.pen_pre), hid, (int64_t)B * 512);
      PilotEncGrads pg;
      for (int e = 0; e < 8; ++e) {
        if (!gemm_dw<4>(dstk + e * 256, 2048, 256, hid + e * 64, 512, 64, B, gr[W_PENC(e, 2)], 64, s, gr[W_PENC(e, 3)]))
          colsum(dstk + e * 256, 2048, B, 256, nullptr, gr[W_PENC(e, 3)], s);
        pg.W2[e] = w[W_PENC(e, 2)]; pg.dw0[e] = gr[W_PENC(e, 0)]; pg.db0[e] = gr[W_PENC(e, 1)];
      }
      hipLaunchKernelGGL(k_pilot_enc_bwd, dim3(B, 8), dim3(64), 0, s, dstk, A(F.pen_pre), properties, pg);
    }
  }

  // ------------------------------------------------------------------ time path (model1.py:705-706, :288)
  {
    float* dstemb = T(L.b256b);
    for (int l = 0; l < NL; ++l) {
      const float* dmod = T(L.dmod[l]);
      if (l == 0)
        gemm_dx<8, MUL_NONE, false>(dmod, 1536, B, 1536, w[W_DIT(l, DIT_ADA_W)], 256, 256, dstemb, 256, nullptr, 0, s);
      else
        gemm_dx<8, MUL_NONE, true>(dmod, 1536, B, 1536, w[W_DIT(l, DIT_ADA_W)], 256, 256, dstemb, 256, nullptr, 0, s);
      if (!gemm_dw<8>(dmod, 1536, 1536, A(F.stemb), 256, 256, B, gr[W_DIT(l, DIT_ADA_W)], 256, s, gr[W_DIT(l, DIT_ADA_B)]))
        colsum(dmod, 1536, B, 1536, nullptr, gr[W_DIT(l, DIT_ADA_B)], s);
    }
    float* dtemb = T(L.b256c);
    hipLaunchKernelGGL(k_time_bwd_elem, dim3(ceil_div(B * 256, 256)), dim3(256), 0, s, dmolsum, dstemb, A(F.temb), dtemb, B * 256);
    if (!gemm_dw<8>(dtemb, 256, 256, A(F.t1), 512, 512, B, gr[W_T2_W], 512, s, gr[W_T2_B]))
      colsum(dtemb, 256, B, 256, nullptr, gr[W_T2_B], s);
    float* dt1 = T(L.b512);
    gemm_dx<8, MUL_GELU_GRAD, false>(dtemb, 256, B, 256, w[W_T2_W], 512, 512, dt1, 512, A(F.t1_pre), 512, s);
    if (!gemm_dw<8>(dt1, 512, 512, A(F.tf), 256, 256, B, gr[W_T0_W], 256, s, gr[W_T0_B]))
      colsum(dt1, 512, B, 512, nullptr, gr[W_T0_B], s);
  }
  side.join();   // the caller's stream now also waits for every weight-gradient launch (and once more at scope exit)
  LD_CHECK_LAUNCH();
  if (!side.ok) return LIGDIFF_ERR_LAUNCH;
  if (!pool_ok) return LIGDIFF_ERR_WORKSPACE;   // (cannot happen while bwd_pool_bytes lists every fresh() above)
  return LIGDIFF_OK;
}


extern "C" int ligdiff_backward(const float* const* w, float* const* gr, const ligdiff_graph* g, const float* x,
                                const float* pos, const float* properties, int flags, const float* d_atom,
                                const float* d_pos, const float* d_prop, const void* fwd_workspace,
                                void* workspace, size_t workspace_bytes, void* stream) {
  return ligdiff_backward_train(w, gr, g, x, pos, properties, flags, d_atom, d_pos, d_prop, fwd_workspace, workspace,
                                workspace_bytes, stream, nullptr);
}

extern "C" size_t ligdiff_backward_workspace_bytes_opt(int64_t n_atoms, int64_t n_edges, int32_t n_mols,
                                                       const ligdiff_options* options) {
  if (check_options(options)) return 0;
  CallOptions scope(options);
  return ligdiff_backward_workspace_bytes(n_atoms, n_edges, n_mols);
}
extern "C" int ligdiff_backward_opt(const float* const* w, float* const* gr, const ligdiff_graph* g, const float* x,
                                    const float* pos, const float* properties, int flags, const float* d_atom,
                                    const float* d_pos, const float* d_prop, const void* fwd_workspace,
                                    void* workspace, size_t workspace_bytes, void* stream,
                                    const ligdiff_dropout* dropout, const ligdiff_options* options) {
  if (check_options(options)) return LIGDIFF_ERR_ARG;
  CallOptions scope(options);
  return ligdiff_backward_train(w, gr, g, x, pos, properties, flags, d_atom, d_pos, d_prop, fwd_workspace, workspace,
                                workspace_bytes, stream, dropout);
}

// Measurement hook for bench.py: runs `reps` backward passes (gradients keep accumulating into
// `gr`: give it throw-away buffers) and returns the mean duration, from HIP events on `stream` net of
// an empty event pair, of the `wide_dw_index`-th gemm_dw_kernel<4> launch of a pass, with its
// shape in dims3_host = {nout, kin, rows} (dW[nout, kin] += dY[rows, nout]^T X[rows, kin]).
extern "C" int ligdiff_backward_timed(const float* const* w, float* const* gr, const ligdiff_graph* g, const float* x,
                                      const float* pos, const float* properties, int flags, const float* d_atom,
                                      const float* d_pos, const float* d_prop, const void* fwd_workspace,
                                      void* workspace, size_t workspace_bytes, void* stream, int32_t wide_dw_index,
                                      int32_t reps, float* mean_ms_host, int64_t* dims3_host) {
  if (wide_dw_index < 0 || reps <= 0 || !mean_ms_host || !dims3_host) return LIGDIFF_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  BwdProbe p;
  memset(&p, 0, sizeof p);
  p.want = wide_dw_index;
  if (hipEventCreate(&p.e0) != hipSuccess || hipEventCreate(&p.e1) != hipSuccess) return LIGDIFF_ERR_LAUNCH;
  int rc = LIGDIFF_OK;
  float empty_ms = 1e30f;
  for (int i = 0; i < 5; ++i) {
    float ms = 0.f;
    if (hipEventRecord(p.e0, s) != hipSuccess || hipEventRecord(p.e1, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess || hipEventElapsedTime(&ms, p.e0, p.e1) != hipSuccess) {
      rc = LIGDIFF_ERR_LAUNCH;
      break;
    }
    empty_ms = ms < empty_ms ? ms : empty_ms;
  }
  double total = 0.0;
  for (int i = 0; i < reps && rc == LIGDIFF_OK; ++i) {
    p.seen = 0; p.hits = 0;
    g_bwd_probe = &p;
    g_bwd_one_stream = true;   // the kernel's own duration: a training step overlaps it with the input-gradient chain
    rc = ligdiff_backward(w, gr, g, x, pos, properties, flags, d_atom, d_pos, d_prop, fwd_workspace, workspace,
                          workspace_bytes, stream);
    g_bwd_one_stream = false;
    g_bwd_probe = nullptr;
    if (rc == LIGDIFF_OK && hipStreamSynchronize(s) != hipSuccess) rc = LIGDIFF_ERR_LAUNCH;
    if (rc == LIGDIFF_OK && p.hits != 1) rc = LIGDIFF_ERR_ARG;   // no such launch in this pass
    float ms = 0.f;
    if (rc == LIGDIFF_OK && hipEventElapsedTime(&ms, p.e0, p.e1) != hipSuccess) rc = LIGDIFF_ERR_LAUNCH;
    total += ms > empty_ms ? ms - empty_ms : 0.f;
  }
  (void)hipEventDestroy(p.e0);
  (void)hipEventDestroy(p.e1);
  *mean_ms_host = (float)(total / reps);
  dims3_host[0] = p.nout; dims3_host[1] = p.kin; dims3_host[2] = p.rows;
  return rc;
}
