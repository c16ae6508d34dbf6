// Host-side launch helpers shared by the forward pass and the consistency module.
#pragma once
#include <stdlib.h>
#include <string.h>

#include "epilogues.h"
#include "gemm3.h"
#include "gemm4.h"
#include "split_cache.h"

// ------------------------------------------------------------------ launch helpers
template <int NT, int WMODE, class AL, class EPI>
static inline void launch_gemm(const GemmCore& g, const AL& al, const EPI& epi, int nb, hipStream_t s) {
  if (g.M <= 0) return;
  dim3 grid(ceil_div(g.M, G_BM), nb);
  hipLaunchKernelGGL((gemm_rows_kernel<NT, WMODE, AL, EPI>), grid, dim3(256), 0, s, g, al, epi);
}

// LDS-DMA main loop (gemm2.h) unless LIGDIFF_GEMM_V1 is set in the environment (A/B switch).
static inline bool use_gemm2() {
  static const bool v = getenv("LIGDIFF_GEMM_V1") == nullptr;
  return v;
}
// Eight-wave column-split kernel (gemm3.h) for 256-column blocks unless LIGDIFF_GEMM_NOCS is set.
static inline bool use_cs() {
  static const bool v = getenv("LIGDIFF_GEMM_NOCS") == nullptr && getenv("LIGDIFF_GEMM_V1") == nullptr;
  return v;
}
// Fused chain kernels (*_chain.h) on the inference path.  LIGDIFF_CHAIN in the environment: "0" never
// (LIGDIFF_NO_CHAIN set means the same), "force" always, otherwise by fill: a chain workgroup (128 rows,
// one wave per SIMD, > 100 KiB of LDS) has a CU to itself for the whole chain, so a launch of n workgroups
// takes ceil(n / CUs) rounds whatever n is - a launch that leaves more than half of the chip empty (a small
// batch) is faster on the separate launches, whose column blocks spread over the chip (29-atom ligands:
// 512 molecules 1.89 against 2.11 ms per reverse step, 600 molecules 2.42 against 2.32; beyond one round
// the chain kernels win at every size measured, 257 workgroups included).
int chain_mode_override();          // ligdiff_set_chain_mode: -1 = as the environment says
static inline int chain_mode() {   // 0 off, 1 by fill, 2 force
  static const int v = [] {
    if (getenv("LIGDIFF_GEMM_NOCS") != nullptr || getenv("LIGDIFF_GEMM_V1") != nullptr || getenv("LIGDIFF_NO_CHAIN") != nullptr) return 0;
    const char* e = getenv("LIGDIFF_CHAIN");
    if (e == nullptr || e[0] == '\0') return 1;
    if (e[0] == '0') return 0;
    return (e[0] == 'f' || e[0] == 'F') ? 2 : 1;
  }();
  const int o = chain_mode_override();
  return o >= 0 ? o : v;
}
// Molecule-per-wave form of the attention-layer kernel (attention inside it) unless LIGDIFF_CHAIN_MOL=0 (A/B switch).
static inline bool chain_mol_allowed() {
  static const bool v = [] { const char* e = getenv("LIGDIFF_CHAIN_MOL"); return !(e != nullptr && e[0] == '0'); }();
  return v;
}
static inline int device_cus() {   // compute units of the current device
  static thread_local int cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cus[dev] == 0 && hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus[dev] = 256;
  return cus[dev] > 0 ? cus[dev] : 256;
}
// Callers that run k independent sub-batches concurrently on k streams of one GPU say so with
// ligdiff_set_concurrent_calls(k): a launch then shares the chip with k - 1 like it, and the fill is judged on k times
// its own workgroups.
int chain_concurrent_calls();
static inline bool use_chain(int64_t rows) {
  const int mode = chain_mode();
  if (mode != 1) return mode == 2;
  const int64_t n = (rows + 127) / 128 * chain_concurrent_calls(), cu = device_cus();
  return n > 0 && 2 * n >= cu;   // at least half of the chip's workgroup slots are used
}
// REDUCE form of the fused edge kernel (edge_chain.h).  LIGDIFF_EDGE_REDUCE in the environment: "0" never, "1"
// whenever the fused edge kernel runs, otherwise for graphs that average at least eight incoming edges per atom (there
// the [E,256] message tensor is the largest buffer of a layer; on bond-template graphs of two or three edges per atom
// the aggregation launch is already a few microseconds of cache reads).
static inline bool edge_reduce_wanted(int64_t n_atoms, int64_t n_edges) {
  static const int v = [] { const char* e = getenv("LIGDIFF_EDGE_REDUCE"); return e == nullptr || e[0] == '\0' ? -1 : e[0] != '0'; }();
  return v < 0 ? n_edges >= 8 * n_atoms : v != 0;
}
// Column-split GEMM launches of few rows: a 128-row workgroup is eight waves, two to a SIMD, and a wave's tile is a
// fixed chain of MFMAs, so a launch whose 128-row workgroups cover at most half of the CUs runs as 64-row
// workgroups (four waves, one to a SIMD, on twice as many CUs) in about half the time (2n <= CUs is the first case
// of the rule below).  Same bits either way
// (LIGDIFF_GEMM_TILE64=0 switches the choice off for A/B runs).
static inline bool half_tiles(int M, int nb) {
  static const bool on = [] { const char* e = getenv("LIGDIFF_GEMM_TILE64"); return !(e != nullptr && e[0] == '0'); }();
  const int64_t n = (int64_t)ceil_div(M, 128) * nb * chain_concurrent_calls(), cu = device_cus();
  // also at larger sizes whenever the finer grain lowers the most waves a SIMD gets: ceil(2n / CUs) four-wave
  // workgroups on the fullest CU against ceil(n / CUs) eight-wave ones (measured: reverse step 1.49 -> 1.44 ms at
  // 400 ligands of 29 atoms, 1.54 -> 1.50 at 512; training step 8.0 -> 7.7 ms at 400, 10.7 -> 10.4 at 600)
  return on && (2 * n + cu - 1) / cu < 2 * ((n + cu - 1) / cu);
}
// Launches of very few rows: with at most one 32-row workgroup per CU the four-wave 32-row form (gemm_cs_quarter_kernel)
// halves the chain of MFMAs a wave runs once more.  Same bits, same switch.
static inline bool quarter_tiles(int M, int nb) {
  static const bool on = [] { const char* e = getenv("LIGDIFF_GEMM_TILE64"); return !(e != nullptr && e[0] == '0'); }();
  const int64_t n = (int64_t)ceil_div(M, 32) * nb * chain_concurrent_calls();
  return on && n <= device_cus();
}
// Split-operand mode (gemm4.h): while a cache is installed for the calling thread, the column-split
// launches below convert each weight block they meet for the first time (one small launch on the
// same stream) and run the bf16-pipe kernel; without a cache, or once its region is full, they run
// the fp32-pipe kernel.
static thread_local SplitCache* g_split = nullptr;

template <int NTW>
static inline bool split_blocks(const GemmCore& g, int nb, hipStream_t s, SplitW* out) {
  SplitCache* c = g_split;
  if (c == nullptr || c->base == nullptr || nb > 8) return false;
  const int ncols = 64 * NTW;
  const size_t need = (size_t)(g.K / G2_BK) * G4_CHUNK_ELEMS(NTW);
  for (int b = 0; b < nb; ++b) {
    const uint16_t* blk = nullptr;
    for (int i = 0; i < c->n; ++i) {
      const SplitCache::Entry& e = c->ent[i];
      if (e.W == g.W[b] && e.ldw == g.ldw && e.K == g.K && e.n_valid == g.n_valid && e.ncols == ncols) {
        blk = e.blk;
        break;
      }
    }
    if (blk == nullptr) {
      if (c->n >= (int)(sizeof c->ent / sizeof c->ent[0]) || c->used + need > c->cap) return false;
      uint16_t* dst = c->base + c->used;
      hipLaunchKernelGGL(k_split_weight, dim3(ceil_div((g.K / G2_BK) * ncols, 256)), dim3(256), 0, s, g.W[b], g.ldw,
                         g.n_valid, g.K, ncols, dst);
      c->ent[c->n++] = SplitCache::Entry{g.W[b], g.ldw, g.K, g.n_valid, ncols, dst};
      c->used += need;
      blk = dst;
    }
    out->blk[b] = blk;
  }
  return true;
}

template <class AP, class EPI>
static inline void launch_gemm_cs(const GemmCore& g, const AP& ap, const EPI& epi, int nb, hipStream_t s) {
  if (g.M <= 0) return;
  SplitW sw;
  if (g_split != nullptr && split_blocks<4>(g, nb, s, &sw)) {
    hipLaunchKernelGGL((gemm_cs_split_kernel<AP, EPI, 4>), dim3(ceil_div(g.M, 128), nb), dim3(512), 0, s, g, sw, ap, epi);
    return;
  }
  if (quarter_tiles(g.M, nb)) {
    hipLaunchKernelGGL((gemm_cs_quarter_kernel<AP, EPI>), dim3(ceil_div(g.M, 32), nb), dim3(256), 0, s, g, ap, epi);
    return;
  }
  if (half_tiles(g.M, nb)) {
    hipLaunchKernelGGL((gemm_cs_kernel<AP, EPI, 4, 2>), dim3(ceil_div(g.M, 64), nb), dim3(256), 0, s, g, ap, epi);
    return;
  }
  hipLaunchKernelGGL((gemm_cs_kernel<AP, EPI>), dim3(ceil_div(g.M, 128), nb), dim3(512), 0, s, g, ap, epi);
}
// Four-wave row-slab launch for epilogues that need whole 128-column rows in one wave (operator() form):
// the split-operand kernel when a cache is installed, else the fp32 LDS-DMA kernel of gemm2.h.
template <class AP, class EPI>
static inline void launch_gemm_rows128(const GemmCore& g, const AP& ap, const EPI& epi, int nb, hipStream_t s) {
  if (g.M <= 0) return;
  SplitW sw;
  if (g_split != nullptr && split_blocks<2>(g, nb, s, &sw)) {
    hipLaunchKernelGGL((gemm_cs_split_kernel<AP, EPI, 4, 2, 4>), dim3(ceil_div(g.M, 128), nb), dim3(256), 0, s, g, sw, ap, epi);
    return;
  }
  hipLaunchKernelGGL((gemm_rows2_kernel<4, AP, EPI>), dim3(ceil_div(g.M, G_BM), nb), dim3(256), 0, s, g, ap, epi);
}
// 128-column blocks (n_valid <= 128 rows of W) on the eight-wave kernel with two accumulators per wave
template <class AP, class EPI>
static inline void launch_gemm_cs128(const GemmCore& g, const AP& ap, const EPI& epi, int nb, hipStream_t s) {
  if (g.M <= 0) return;
  SplitW sw;
  if (g_split != nullptr && split_blocks<2>(g, nb, s, &sw)) {
    hipLaunchKernelGGL((gemm_cs_split_kernel<AP, EPI, 2>), dim3(ceil_div(g.M, 128), nb), dim3(512), 0, s, g, sw, ap, epi);
    return;
  }
  if (half_tiles(g.M, nb)) {
    hipLaunchKernelGGL((gemm_cs_kernel<AP, EPI, 2, 2>), dim3(ceil_div(g.M, 64), nb), dim3(256), 0, s, g, ap, epi);
    return;
  }
  hipLaunchKernelGGL((gemm_cs_kernel<AP, EPI, 2>), dim3(ceil_div(g.M, 128), nb), dim3(512), 0, s, g, ap, epi);
}
template <int NT, class EPI>
static inline void launch_gemm_plain(const GemmCore& g, const float* A, int lda, const EPI& epi, int nb, hipStream_t s) {
  if (g.M <= 0) return;
  dim3 grid(ceil_div(g.M, G_BM), nb);
  if constexpr (NT == 4) {
    if (use_cs() && (g.K % G2_BK) == 0 && (lda % 4) == 0 && (g.ldw % 4) == 0 && g.n_valid <= 128) {
      launch_gemm_cs128(g, APtrPlain{A, lda}, epi, nb, s);
      return;
    }
  }
  if constexpr (NT == 8) {
    if (use_cs() && (g.K % G2_BK) == 0 && (lda % 4) == 0 && (g.ldw % 4) == 0 && g.n_valid == 256) {
      launch_gemm_cs(g, APtrPlain{A, lda}, epi, nb, s);
      return;
    }
  }
  if (use_gemm2() && (g.K % G2_BK) == 0 && (lda % 4) == 0 && (g.ldw % 4) == 0) {
    APtrPlain ap{A, lda};
    hipLaunchKernelGGL((gemm_rows2_kernel<NT, APtrPlain, EPI>), grid, dim3(256), 0, s, g, ap, epi);
  } else {
    ALoadPlain al{A, lda, g.M, g.K};
    hipLaunchKernelGGL((gemm_rows_kernel<NT, W_ALIGNED, ALoadPlain, EPI>), grid, dim3(256), 0, s, g, al, epi);
  }
}

static inline GemmCore core_blocks(int M, int K, int n_valid, const float* W, int ldw, int nb, int rows_per_block) {
  GemmCore g;
  memset(&g, 0, sizeof g);
  g.M = M; g.K = K; g.n_valid = n_valid; g.ldw = ldw;
  for (int i = 0; i < nb && i < 8; ++i) g.W[i] = W + (size_t)i * rows_per_block * ldw;
  return g;
}

// y[M, nb*256] = act(A[M,K] W^T + b).  Plain bias/activation epilogues need no full rows, so
// they run as 128-column blocks: twice the workgroups with half the accumulators each, which
// keeps four or more workgroups resident per CU and lets one's epilogue (activation + stores)
// overlap another's MFMA loop.
template <int ACT, bool DUAL>
static inline void gemm_bias_act(const float* A, int lda, int M, int K, const float* W, int ldw, const float* bias,
                          int nb, float* C, int ldc, float* C2, hipStream_t s, float* pre_out = nullptr) {
  const int nb4 = nb * 2;
  if (use_cs() && (K % G2_BK) == 0 && (lda % 4) == 0 && (ldw % 4) == 0) {
    GemmCore g = core_blocks(M, K, 256, W, ldw, nb, 256);
    EpiBiasAct<8, ACT, DUAL> e;
    memset(&e, 0, sizeof e);
    e.M = M; e.n_valid = 256; e.col_block = 256; e.C = C; e.ldc = ldc; e.C2 = C2; e.alpha = 1.0f;
    e.pre_out = pre_out;
    for (int i = 0; i < nb; ++i) e.bias[i] = bias ? bias + (size_t)i * 256 : nullptr;
    launch_gemm_cs(g, APtrPlain{A, lda}, e, nb, s);
    return;
  }
  if (nb4 <= 8 && use_gemm2()) {
    GemmCore g = core_blocks(M, K, 128, W, ldw, nb4, 128);
    EpiBiasAct<4, ACT, DUAL> e;
    memset(&e, 0, sizeof e);
    e.M = M; e.n_valid = 128; e.col_block = 128; e.C = C; e.ldc = ldc; e.C2 = C2; e.alpha = 1.0f;
    e.pre_out = pre_out;
    for (int i = 0; i < nb4; ++i) e.bias[i] = bias ? bias + (size_t)i * 128 : nullptr;
    launch_gemm_plain<4>(g, A, lda, e, nb4, s);
    return;
  }
  GemmCore g = core_blocks(M, K, 256, W, ldw, nb, 256);
  EpiBiasAct<8, ACT, DUAL> e;
  memset(&e, 0, sizeof e);
  e.M = M; e.n_valid = 256; e.col_block = 256; e.C = C; e.ldc = ldc; e.C2 = C2; e.alpha = 1.0f;
  e.pre_out = pre_out;
  for (int i = 0; i < nb; ++i) e.bias[i] = bias ? bias + (size_t)i * 256 : nullptr;
  launch_gemm_plain<8>(g, A, lda, e, nb, s);
}

