// Score-network forward: host-side launch sequence + workspace layout.
#include <stdio.h>
#include <string.h>

#include <stdlib.h>

#include "epilogues.h"
#include "gemm3.h"
#include "chain_args.h"
#include "forward.h"
#include "kernels.h"
#include "weights.h"

// ------------------------------------------------------------------ weights table
static void wt_set(WeightInfo* t, int slot, int64_t numel, const char* fmt, int a = 0, int b = 0) {
  snprintf(t[slot].name, sizeof(t[slot].name), fmt, a, b);
  t[slot].numel = numel;
}
const WeightInfo* weight_table() {
  static WeightInfo tab[W_COUNT];
  static const bool once = [] {
    WeightInfo* t = tab;
    wt_set(t, W_ATOM_W, 256 * 119, "atom_embedding.0.weight");
    wt_set(t, W_ATOM_B, 256, "atom_embedding.0.bias");
    wt_set(t, W_ATOM_G, 256, "atom_embedding.1.weight");
    wt_set(t, W_ATOM_BE, 256, "atom_embedding.1.bias");
    wt_set(t, W_POS_W, 256 * 3, "pos_embedding.0.weight");
    wt_set(t, W_POS_B, 256, "pos_embedding.0.bias");
    wt_set(t, W_POS_G, 256, "pos_embedding.1.weight");
    wt_set(t, W_POS_BE, 256, "pos_embedding.1.bias");
    wt_set(t, W_T0_W, 512 * 256, "time_mlp.0.weight");
    wt_set(t, W_T0_B, 512, "time_mlp.0.bias");
    wt_set(t, W_T2_W, 256 * 512, "time_mlp.2.weight");
    wt_set(t, W_T2_B, 256, "time_mlp.2.bias");
    wt_set(t, W_NULL_EMB, 256, "property_guidance.null_embedding");
    wt_set(t, W_PAR0_W, 256 * 2048, "property_guidance.pareto_predictor.0.weight");
    wt_set(t, W_PAR0_B, 256, "property_guidance.pareto_predictor.0.bias");
    wt_set(t, W_PAR2_W, 8 * 256, "property_guidance.pareto_predictor.2.weight");
    wt_set(t, W_PAR2_B, 8, "property_guidance.pareto_predictor.2.bias");
    for (int i = 0; i < 8; ++i) {
      wt_set(t, W_PENC(i, 0), 64, "property_guidance.property_encoders.%d.0.weight", i);
      wt_set(t, W_PENC(i, 1), 64, "property_guidance.property_encoders.%d.0.bias", i);
      wt_set(t, W_PENC(i, 2), 256 * 64, "property_guidance.property_encoders.%d.2.weight", i);
      wt_set(t, W_PENC(i, 3), 256, "property_guidance.property_encoders.%d.2.bias", i);
    }
    for (int l = 0; l < LD_MAXL; ++l) {
      wt_set(t, W_E3(l, E3_RB0_W), 32, "equivariant_layers.%d.radial_basis.0.weight", l);
      wt_set(t, W_E3(l, E3_RB0_B), 32, "equivariant_layers.%d.radial_basis.0.bias", l);
      wt_set(t, W_E3(l, E3_RB2_W), 32 * 32, "equivariant_layers.%d.radial_basis.2.weight", l);
      wt_set(t, W_E3(l, E3_RB2_B), 32, "equivariant_layers.%d.radial_basis.2.bias", l);
      wt_set(t, W_E3(l, E3_EN0_W), 256 * 544, "equivariant_layers.%d.edge_network.0.weight", l);
      wt_set(t, W_E3(l, E3_EN0_B), 256, "equivariant_layers.%d.edge_network.0.bias", l);
      wt_set(t, W_E3(l, E3_EN2_W), 256 * 256, "equivariant_layers.%d.edge_network.2.weight", l);
      wt_set(t, W_E3(l, E3_EN2_B), 256, "equivariant_layers.%d.edge_network.2.bias", l);
      wt_set(t, W_E3(l, E3_EN3_G), 256, "equivariant_layers.%d.edge_network.3.weight", l);
      wt_set(t, W_E3(l, E3_EN3_B), 256, "equivariant_layers.%d.edge_network.3.bias", l);
      wt_set(t, W_E3(l, E3_NN0_W), 256 * 512, "equivariant_layers.%d.node_network.0.weight", l);
      wt_set(t, W_E3(l, E3_NN0_B), 256, "equivariant_layers.%d.node_network.0.bias", l);
      wt_set(t, W_E3(l, E3_NN2_W), 256 * 256, "equivariant_layers.%d.node_network.2.weight", l);
      wt_set(t, W_E3(l, E3_NN2_B), 256, "equivariant_layers.%d.node_network.2.bias", l);
      wt_set(t, W_E3(l, E3_NN3_G), 256, "equivariant_layers.%d.node_network.3.weight", l);
      wt_set(t, W_E3(l, E3_NN3_B), 256, "equivariant_layers.%d.node_network.3.bias", l);
      wt_set(t, W_E3(l, E3_CN0_W), 128 * 256, "equivariant_layers.%d.coord_network.0.weight", l);
      wt_set(t, W_E3(l, E3_CN0_B), 128, "equivariant_layers.%d.coord_network.0.bias", l);
      wt_set(t, W_E3(l, E3_CN2_W), 128, "equivariant_layers.%d.coord_network.2.weight", l);
      wt_set(t, W_E3(l, E3_CN2_B), 1, "equivariant_layers.%d.coord_network.2.bias", l);
    }
    for (int l = 0; l < LD_MAXL; ++l) {
      wt_set(t, W_DIT(l, DIT_ADA_W), 1536 * 256, "dit_layers.%d.adaLN_modulation.1.weight", l);
      wt_set(t, W_DIT(l, DIT_ADA_B), 1536, "dit_layers.%d.adaLN_modulation.1.bias", l);
      wt_set(t, W_DIT(l, DIT_INP_W), 768 * 256, "dit_layers.%d.attention.in_proj_weight", l);
      wt_set(t, W_DIT(l, DIT_INP_B), 768, "dit_layers.%d.attention.in_proj_bias", l);
      wt_set(t, W_DIT(l, DIT_OUT_W), 256 * 256, "dit_layers.%d.attention.out_proj.weight", l);
      wt_set(t, W_DIT(l, DIT_OUT_B), 256, "dit_layers.%d.attention.out_proj.bias", l);
      wt_set(t, W_DIT(l, DIT_FF0_W), 1024 * 256, "dit_layers.%d.ff_network.0.weight", l);
      wt_set(t, W_DIT(l, DIT_FF0_B), 1024, "dit_layers.%d.ff_network.0.bias", l);
      wt_set(t, W_DIT(l, DIT_FF3_W), 256 * 1024, "dit_layers.%d.ff_network.3.weight", l);
      wt_set(t, W_DIT(l, DIT_FF3_B), 256, "dit_layers.%d.ff_network.3.bias", l);
    }
    const char* heads[3] = {"atom_output", "pos_output", "property_output"};
    const int outs[3] = {119, 3, 15};
    for (int h = 0; h < 3; ++h) {
      char f[96];
      snprintf(f, sizeof f, "%s.0.weight", heads[h]); wt_set(t, W_HEAD(h, HEAD_W0), 256 * 256, f);
      snprintf(f, sizeof f, "%s.0.bias", heads[h]);   wt_set(t, W_HEAD(h, HEAD_B0), 256, f);
      snprintf(f, sizeof f, "%s.2.weight", heads[h]); wt_set(t, W_HEAD(h, HEAD_G), 256, f);
      snprintf(f, sizeof f, "%s.2.bias", heads[h]);   wt_set(t, W_HEAD(h, HEAD_BE), 256, f);
      snprintf(f, sizeof f, "%s.3.weight", heads[h]); wt_set(t, W_HEAD(h, HEAD_W3), outs[h] * 256, f);
      snprintf(f, sizeof f, "%s.3.bias", heads[h]);   wt_set(t, W_HEAD(h, HEAD_B3), outs[h], f);
    }
    wt_set(t, W_TIME_FREQS, 128, "__time_freqs__");
    {
      const char* pred[3] = {"atom_predictor", "bond_predictor", "valency_predictor"};
      const int in0[3] = {256, 516, 375}, last_layer[3] = {6, 6, 5}, last_out[3] = {119, 5, 1};
      for (int p = 0; p < 3; ++p) {
        char f[96];
        snprintf(f, sizeof f, "moldiff_consistency.%s.0.weight", pred[p]); wt_set(t, W_CONS(p, CONS_W0), 256 * in0[p], f);
        snprintf(f, sizeof f, "moldiff_consistency.%s.0.bias", pred[p]);   wt_set(t, W_CONS(p, CONS_B0), 256, f);
        snprintf(f, sizeof f, "moldiff_consistency.%s.2.weight", pred[p]); wt_set(t, W_CONS(p, CONS_G), 256, f);
        snprintf(f, sizeof f, "moldiff_consistency.%s.2.bias", pred[p]);   wt_set(t, W_CONS(p, CONS_BE), 256, f);
        snprintf(f, sizeof f, "moldiff_consistency.%s.3.weight", pred[p]); wt_set(t, W_CONS(p, CONS_W3), 128 * 256, f);
        snprintf(f, sizeof f, "moldiff_consistency.%s.3.bias", pred[p]);   wt_set(t, W_CONS(p, CONS_B3), 128, f);
        snprintf(f, sizeof f, "moldiff_consistency.%s.%d.weight", pred[p], last_layer[p]);
        wt_set(t, W_CONS(p, CONS_WL), last_out[p] * 128, f);
        snprintf(f, sizeof f, "moldiff_consistency.%s.%d.bias", pred[p], last_layer[p]);
        wt_set(t, W_CONS(p, CONS_BL), last_out[p], f);
      }
      wt_set(t, W_CONS_MAXVAL, 119, "moldiff_consistency.max_valences");
      wt_set(t, W_CONS_RADII, 119, "moldiff_consistency.covalent_radii");
    }
    return true;
  }();
  (void)once;
  return tab;
}

#include "launch.h"

// ------------------------------------------------------------------ per-call options and process-wide defaults
#include <atomic>
static thread_local const ligdiff_options* t_options = nullptr;
CallOptions::CallOptions(const ligdiff_options* o) : prev(t_options) { t_options = o; }
CallOptions::~CallOptions() { t_options = prev; }
const ligdiff_options* call_options() { return t_options; }
extern "C" void ligdiff_options_init(ligdiff_options* o) {
  if (o == nullptr) return;
  o->gemm_mode = -1; o->chain_mode = -1; o->concurrent_calls = 0;
  o->edge_chain = -1; o->node_chain = -1; o->mol_wave = -1; o->edge_reduce = -1; o->cfg_two_pass = -1;
  o->layer_pairs = -1;
}
int call_layer_pairs() { return (t_options != nullptr && t_options->layer_pairs > 0) ? t_options->layer_pairs : 3; }
int check_options(const ligdiff_options* o) {
  if (o == nullptr) return LIGDIFF_OK;
  auto tri = [](int v) { return v >= -1 && v <= 1; };
  if (!tri(o->gemm_mode) || o->chain_mode < -1 || o->chain_mode > 2 || o->concurrent_calls < 0 || o->concurrent_calls > 16 ||
      !tri(o->edge_chain) || !tri(o->node_chain) || !tri(o->mol_wave) || !tri(o->edge_reduce) || !tri(o->cfg_two_pass) ||
      o->layer_pairs < -1 || o->layer_pairs > LD_MAXL)
    return LIGDIFF_ERR_ARG;
  return LIGDIFF_OK;
}
static std::atomic<int> g_split_mode{-1};   // -1: not decided yet (environment), 0 / 1
bool split_gemm_enabled() {
  if (t_options != nullptr && t_options->gemm_mode >= 0) return t_options->gemm_mode != 0;
  int m = g_split_mode.load(std::memory_order_relaxed);
  if (m < 0) {
    const char* e = getenv("LIGDIFF_GEMM_SPLIT");
    m = (e != nullptr && e[0] != '\0' && e[0] != '0') ? 1 : 0;
    g_split_mode.store(m, std::memory_order_relaxed);
  }
  return m != 0;
}
bool fwd_column_split_gemms() { return use_cs(); }
bool cfg_one_pass() {   // the environment is read per call: the tests flip it within one process
  if (!use_cs()) return false;
  if (t_options != nullptr && t_options->cfg_two_pass >= 0) return t_options->cfg_two_pass == 0;
  return getenv("LIGDIFF_CFG_TWO_PASS") == nullptr;
}
static std::atomic<int> g_chain_mode{-1};
int chain_mode_override() {
  if (t_options != nullptr && t_options->chain_mode >= 0) return t_options->chain_mode;
  return g_chain_mode.load(std::memory_order_relaxed);
}
extern "C" int ligdiff_set_chain_mode(int mode) {
  if (mode < -1 || mode > 2) return -2;
  return g_chain_mode.exchange(mode, std::memory_order_relaxed);
}
static std::atomic<int> g_concurrent_calls{1};
int chain_concurrent_calls() {
  if (t_options != nullptr && t_options->concurrent_calls > 0) return t_options->concurrent_calls;
  return g_concurrent_calls.load(std::memory_order_relaxed);
}
extern "C" int ligdiff_set_concurrent_calls(int k) {
  if (k < 1 || k > 16) return -1;
  return g_concurrent_calls.exchange(k, std::memory_order_relaxed);
}
extern "C" int ligdiff_set_split_gemm(int enabled) {
  int m = g_split_mode.load(std::memory_order_relaxed);
  if (m < 0) {   // the process default, whatever options the calling thread may have installed
    const char* e = getenv("LIGDIFF_GEMM_SPLIT");
    m = (e != nullptr && e[0] != '\0' && e[0] != '0') ? 1 : 0;
  }
  g_split_mode.store(enabled ? 1 : 0, std::memory_order_relaxed);
  return m;
}
extern "C" void ligdiff_default_options(ligdiff_options* o) {
  if (o == nullptr) return;
  ligdiff_options_init(o);
  const ligdiff_options* saved = t_options;
  t_options = nullptr;   // the process-wide defaults, not the calling thread's current call
  o->gemm_mode = split_gemm_enabled() ? 1 : 0;
  o->chain_mode = chain_mode();
  o->concurrent_calls = chain_concurrent_calls();
  o->cfg_two_pass = cfg_one_pass() ? 0 : 1;
  t_options = saved;
}
FwdPlan forward_plan(int64_t N, int64_t E, int64_t B, int max_mol_atoms) {
  const ligdiff_options* o = t_options;
  auto pick = [](int field, bool by_rule) { return field >= 0 ? field != 0 : by_rule; };
  FwdPlan p;
  p.edge_chain = pick(o ? o->edge_chain : -1, use_chain(E));
  p.node_chain = pick(o ? o->node_chain : -1, use_chain(N));
  // molecules of at most 32 atoms that fill at least three quarters of a 32-lane slab each
  p.mol_wave = pick(o ? o->mol_wave : -1, chain_mol_allowed() && max_mol_atoms <= 32 && 4 * N >= 3 * 32 * B && use_chain(32 * B));
  p.edge_reduce = pick(o ? o->edge_reduce : -1, edge_reduce_wanted(N, E));
  return p;
}
extern "C" int ligdiff_plan(int64_t n_atoms, int64_t n_edges, int32_t n_mols, int32_t max_mol_atoms,
                            const ligdiff_options* in, ligdiff_options* out) {
  if (out == nullptr || n_atoms <= 0 || n_edges < 0 || n_mols <= 0 || max_mol_atoms <= 0 || check_options(in)) return LIGDIFF_ERR_ARG;
  ligdiff_options r;
  if (in != nullptr) r = *in; else ligdiff_options_init(&r);
  {
    CallOptions scope(&r);
    const FwdPlan p = forward_plan(n_atoms, n_edges, n_mols, max_mol_atoms);
    r.edge_chain = p.edge_chain; r.node_chain = p.node_chain; r.mol_wave = p.mol_wave; r.edge_reduce = p.edge_reduce;
  }
  *out = r;
  return LIGDIFF_OK;
}
// Bytes reserved for the split weight copies: 6 bytes per weight element plus room for blocks whose
// rows are zero-extended to the 128 / 256 columns of a block (a full region falls back to the fp32 pipe).
static size_t split_region_bytes(int nl) {
  size_t elems = 0;
  for (int i = 0; i < W_CONS_0; ++i)
    if (weight_slot_layer(i) < nl) elems += (size_t)weight_table()[i].numel;
  return align_up(elems * 6 + elems * 6 / 4 + (2u << 20), 256);
}

// ------------------------------------------------------------------ dropout sites
bool make_drop_site(const ligdiff_dropout* d, int layer, int kind, DropSite* out) {
  memset(out, 0, sizeof *out);
  if (d == nullptr) return true;
  const float p = kind == LIGDIFF_DROP_ATTENTION ? d->p_attention[layer]
                  : kind == LIGDIFF_DROP_FFN_HIDDEN ? d->p_ffn_hidden[layer]
                  : kind == LIGDIFF_DROP_FFN_OUT ? d->p_ffn_out[layer]
                  : d->p_consistency[kind - LIGDIFF_DROP_CONS_ATOM];
  if (!(p >= 0.0f && p < 1.0f)) return false;
  if (p == 0.0f) return true;
  out->seed = d->seed;
  out->row_offset = kind == LIGDIFF_DROP_CONS_BOND ? 0 : d->atom_offset;   // pairs are numbered per call
  // sites 0..11: layers 0..2; 15, 16: the consistency module; layers 3.. follow from 32 on
  out->site = kind >= LIGDIFF_DROP_CONS_ATOM ? (uint32_t)(12 + kind)
              : layer < 3 ? (uint32_t)(layer * 4 + kind) : (uint32_t)(32 + layer * 4 + kind);
  const double t = (double)p * 4294967296.0;
  out->thr = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
  if (out->thr == 0u) out->thr = 1u;
  out->inv_keep = 1.0f / (1.0f - p);
  return true;
}
int check_dropout(const ligdiff_dropout* d) {
  if (d == nullptr) return LIGDIFF_OK;
  if (d->atom_offset < 0) return LIGDIFF_ERR_ARG;
  bool any = false;
  for (int l = 0; l < call_layer_pairs(); ++l)
    for (int k = 0; k < 3; ++k) {
      DropSite ds;
      if (!make_drop_site(d, l, k, &ds)) return LIGDIFF_ERR_ARG;
      any = any || ds.thr != 0u;
    }
  for (int k = LIGDIFF_DROP_CONS_ATOM; k <= LIGDIFF_DROP_CONS_BOND; ++k) {
    DropSite ds;
    if (!make_drop_site(d, 0, k, &ds)) return LIGDIFF_ERR_ARG;
  }
  if (any && !use_cs()) return LIGDIFF_ERR_ARG;   // only the column-split GEMM epilogues carry the masks
  return LIGDIFF_OK;
}

__global__ void k_dropout_mask(DropSite d, int kind, int64_t n_atoms, float* __restrict__ out) {
  const int width = kind >= LIGDIFF_DROP_CONS_ATOM ? 128 : kind == LIGDIFF_DROP_FFN_OUT ? 256 : 1024;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_atoms * width) return;
  const int row = (int)(i / width), col = (int)(i % width);
  float v = 1.0f;
  if (d.thr != 0u) {
    if (kind == LIGDIFF_DROP_ATTENTION) {
      const int head = col >> 7, j = col & 127;
      v = drop_scale4(d, drop_att_quad(d, row, head, j))[j & 3];
    } else {
      v = drop_row_scale(d, row, col, width);
    }
  }
  out[i] = v;
}
extern "C" int ligdiff_dropout_mask(const ligdiff_dropout* dropout, int32_t layer, int32_t kind, int64_t n_atoms,
                                    float* out, void* stream) {
  if (!dropout || layer < 0 || layer >= LD_MAXL || kind < 0 || kind > LIGDIFF_DROP_CONS_BOND || n_atoms <= 0 || !out ||
      n_atoms >= (1LL << 30))
    return LIGDIFF_ERR_ARG;
  DropSite ds;
  if (!make_drop_site(dropout, layer, kind, &ds)) return LIGDIFF_ERR_ARG;
  const int width = kind >= LIGDIFF_DROP_CONS_ATOM ? 128 : kind == LIGDIFF_DROP_FFN_OUT ? 256 : 1024;
  hipLaunchKernelGGL(k_dropout_mask, dim3(ceil_div(n_atoms * width, 256)), dim3(256), 0, (hipStream_t)stream, ds, kind,
                     n_atoms, out);
  LD_CHECK_LAUNCH();
  return LIGDIFF_OK;
}

// ------------------------------------------------------------------ workspace layout
void fwd_layout(FwdLayout* L, int64_t N, int64_t E, int32_t B, int flags) {
  memset(L, 0, sizeof *L);
  const bool bcast = (flags & LIGDIFF_FWD_T_BROADCAST) != 0;
  const bool save = (flags & LIGDIFF_FWD_SAVE_FOR_BACKWARD) != 0;
  const int64_t Bt = bcast ? 1 : B;
  const int nl = call_layer_pairs();
  L->nl = nl;
  size_t off = 0;
  auto take = [&](size_t nfloats) { size_t o = off; off += align_up(nfloats * 4, 256); return o; };
  L->Bt = Bt;
  L->tf = take(Bt * 256); L->t1 = take(Bt * 512); L->temb = take(Bt * 256); L->stemb = take(Bt * 256);
  for (int l = 0; l < nl; ++l) L->mod[l] = take(Bt * 1536);
  L->pemb = take((size_t)B * 256); L->pstack = take((size_t)B * 2048); L->pg = take((size_t)B * 256);
  L->h0 = take(N * 256);
  L->xpad = take(N * 128); L->wpad = take(256 * 128);
  L->pos_upd = take(N * 3); L->pos_cur = take(N * 3);
  // Without SAVE every layer reuses layer 0's scratch and the residual stream is updated
  // in place (each element is read and rewritten by the same thread of the epilogue).
  auto per_layer = [&](size_t* arr, size_t nfloats) {
    for (int l = 0; l < nl; ++l) arr[l] = (save || l == 0) ? take(nfloats) : arr[0];
  };
  auto stream_buf = [&](size_t* arr) {
    for (int l = 0; l < nl; ++l) arr[l] = save ? take(N * 256) : L->h0;
  };
  per_layer(L->P, N * 512); per_layer(L->rel, E * 4); per_layer(L->rbf, E * 32);
  per_layer(L->a1, E * 256); per_layer(L->msg, E * 256); per_layer(L->wrel, E * 4);
  per_layer(L->agg, N * 256); per_layer(L->hid, N * 256);
  per_layer(L->xn1, N * 256); per_layer(L->qkv, N * 768); per_layer(L->att, N * 256);
  per_layer(L->xn2, N * 256); per_layer(L->ffh, N * 1024);
  stream_buf(L->h_e3); stream_buf(L->x1); stream_buf(L->x2);
  L->n_slab_cap = ((E + 31) / 32 + B + 3) / 4 * 4;
  L->mol_e0 = take((size_t)B + 1); L->mol_slab0 = take((size_t)B + 1);
  L->slab_e0 = take((size_t)L->n_slab_cap); L->slab_cnt = take((size_t)L->n_slab_cap);
  L->yh = take(N * 512);
  L->pooled = take((size_t)B * 256); L->ph = take((size_t)B * 256);
  L->save = save ? 1 : 0;
  if (save) {
    L->t1_pre = take(Bt * 512); L->ua = take(N * 256);
    L->pen_pre = take((size_t)B * 512); L->pg_pre = take((size_t)B * 256);
    L->pz = take((size_t)B * 8); L->pw = take((size_t)B * 8);
    for (int l = 0; l < nl; ++l) {
      L->pre_e[l] = take(E * 256); L->u_m[l] = take(E * 256); L->hc_pre[l] = take(E * 128);
      L->cw[l] = take(E); L->dist[l] = take(E); L->r1pre[l] = take(E * 32);
      L->hid_pre[l] = take(N * 256); L->u_h[l] = take(N * 256);
      L->oproj[l] = take(N * 256); L->fproj[l] = take(N * 256); L->ffh_pre[l] = take(N * 1024);
      L->lse[l] = take(N * 8);
    }
    L->u_head = take(N * 512); L->u_ph = take((size_t)B * 256);
  }
  L->wsplit = off;
  L->wsplit_bytes = split_gemm_enabled() ? split_region_bytes(nl) : 0;
  off += L->wsplit_bytes;
  L->total = off;
}

// ------------------------------------------------------------------ time path for a table of timesteps
int run_time_rows(const float* const* w, const float* t, int n_rows, float* tf, float* t1, float* temb,
                  float* stemb, float* const* mods, int n_layers, hipStream_t s) {
  if (n_rows <= 0 || n_layers <= 0 || n_layers > LD_MAXL) return LIGDIFF_ERR_ARG;
  hipLaunchKernelGGL(k_fourier, dim3(n_rows), dim3(128), 0, s, t, (const int32_t*)nullptr, w[W_TIME_FREQS], tf);
  SmallLinearSets p0 = {{w[W_T0_W]}, {w[W_T0_B]}, {t1}};
  hipLaunchKernelGGL((k_small_linear<1, false>), dim3(128, 1), dim3(256), 0, s, tf, p0, (float*)nullptr, n_rows, 256, 512);
  SmallLinearSets p2 = {{w[W_T2_W]}, {w[W_T2_B]}, {temb}};
  hipLaunchKernelGGL((k_small_linear<0, true>), dim3(64, 1), dim3(256), 0, s, t1, p2, stemb, n_rows, 512, 256);
  SmallLinearSets pa;
  memset(&pa, 0, sizeof pa);
  for (int l = 0; l < n_layers; ++l) { pa.W[l] = w[W_DIT(l, DIT_ADA_W)]; pa.b[l] = w[W_DIT(l, DIT_ADA_B)]; pa.C[l] = mods[l]; }
  hipLaunchKernelGGL((k_small_linear<0, false>), dim3(384, n_layers), dim3(256), 0, s, stemb, pa, (float*)nullptr, n_rows, 256, 1536);
  LD_CHECK_LAUNCH();
  return LIGDIFF_OK;
}

// ------------------------------------------------------------------ the forward pass
int run_forward(const float* const* w, const ligdiff_graph* g, const float* x, const float* pos,
                const float* t, const int32_t* t_index_dev, const float* properties, int flags,
                float* atom_pred, float* pos_pred, float* prop_pred, char* ws, hipStream_t s, FwdProbe* probe,
                SplitCache* split, const ligdiff_dropout* drop) {
  const int N = (int)g->n_atoms, E = (int)g->n_edges, B = g->n_mols;
  // PB(label) ... PE(): brackets one launch with the probe's events when its label matches
  char lbl[48];
  bool hit = false;
  auto PB = [&](const char* fmt, int l = 0) {
    if (probe == nullptr) return;
    snprintf(lbl, sizeof lbl, fmt, l);
    hit = strcmp(lbl, probe->label) == 0;
    if (hit) (void)hipEventRecord(probe->e0, s);
  };
  auto PE = [&]() {
    if (hit) { (void)hipEventRecord(probe->e1, s); probe->hits++; hit = false; }
  };
  FwdLayout L;
  fwd_layout(&L, N, E, B, flags);
  const int NL = L.nl;   // E(3) layers = attention layers of this model
  for (int i = 0; i < W_CONS_0; ++i)
    if (weight_slot_layer(i) < NL && w[i] == nullptr) return LIGDIFF_ERR_ARG;
  // split-operand mode: the caller's cache (kept over the calls of one sampling loop) or one for this call
  SplitCache local_split;
  struct SplitScope {
    ~SplitScope() { g_split = nullptr; }
  } split_scope;
  if (L.wsplit_bytes > 0) {
    SplitCache* c = split != nullptr ? split : &local_split;
    if (c->base == nullptr) {
      c->base = reinterpret_cast<uint16_t*>(ws + L.wsplit);
      c->cap = L.wsplit_bytes / 2;
    }
    g_split = c;
  }
  auto F = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
  auto S = [&](size_t off) { return L.save ? reinterpret_cast<float*>(ws + off) : (float*)nullptr; };
  const int Bt = (int)L.Bt;
  const int tstride = (flags & LIGDIFF_FWD_T_BROADCAST) ? 0 : 256;
  const int mstride = (flags & LIGDIFF_FWD_T_BROADCAST) ? 0 : 1536;

  // --- time embedding (model1.py:705-706) and adaLN parameters (model1.py:288)
  if (flags & FWD_INTERNAL_TIME_GIVEN) {
    if (!(flags & LIGDIFF_FWD_T_BROADCAST) || L.save) return LIGDIFF_ERR_ARG;
  } else if (Bt <= 4 && !L.save) {  // shared step index: a few rows -> wave-per-output linears
    PB("time.fourier");
    hipLaunchKernelGGL(k_fourier, dim3(Bt), dim3(128), 0, s, t, t_index_dev, w[W_TIME_FREQS], F(L.tf));
    PE();
    SmallLinearSets p0 = {{w[W_T0_W]}, {w[W_T0_B]}, {F(L.t1)}};
    PB("time.mlp0");
    hipLaunchKernelGGL((k_small_linear<1, false>), dim3(128, 1), dim3(256), 0, s, F(L.tf), p0, (float*)nullptr, Bt, 256, 512);
    PE();
    SmallLinearSets p2 = {{w[W_T2_W]}, {w[W_T2_B]}, {F(L.temb)}};
    PB("time.mlp2");
    hipLaunchKernelGGL((k_small_linear<0, true>), dim3(64, 1), dim3(256), 0, s, F(L.t1), p2, F(L.stemb), Bt, 512, 256);
    PE();
    SmallLinearSets pa;
    memset(&pa, 0, sizeof pa);
    for (int l = 0; l < NL; ++l) { pa.W[l] = w[W_DIT(l, DIT_ADA_W)]; pa.b[l] = w[W_DIT(l, DIT_ADA_B)]; pa.C[l] = F(L.mod[l]); }
    PB("dit.adaln");
    hipLaunchKernelGGL((k_small_linear<0, false>), dim3(384, NL), dim3(256), 0, s, F(L.stemb), pa, (float*)nullptr, Bt, 256, 1536);
    PE();
  } else {
    PB("time.fourier");
    hipLaunchKernelGGL(k_fourier, dim3(Bt), dim3(128), 0, s, t, t_index_dev, w[W_TIME_FREQS], F(L.tf));
    PE();
    PB("time.mlp0");
    gemm_bias_act<ACT_GELU, false>(F(L.tf), 256, Bt, 256, w[W_T0_W], 256, w[W_T0_B], 2, F(L.t1), 512, nullptr, s,
                                   S(L.t1_pre));
    PE();
    PB("time.mlp2");
    gemm_bias_act<ACT_NONE, true>(F(L.t1), 512, Bt, 512, w[W_T2_W], 512, w[W_T2_B], 1, F(L.temb), 256, F(L.stemb), s);
    PE();
    for (int l = 0; l < NL; ++l) {
      PB("dit%d.adaln", l);
      gemm_bias_act<ACT_NONE, false>(F(L.stemb), 256, Bt, 256, w[W_DIT(l, DIT_ADA_W)], 256, w[W_DIT(l, DIT_ADA_B)], 6,
                                     F(L.mod[l]), 1536, nullptr, s);
      PE();
    }
  }

  // --- PILOT conditioning (model1.py:714-718, :335-359)
  const float* pemb = nullptr;
  int pstride = 0;
  int pemb_mols = B;   // molecules that receive the property embedding
  if (flags & FWD_INTERNAL_HALF_COND) {
    if (!(flags & LIGDIFF_FWD_PROPERTIES) || (flags & LIGDIFF_FWD_NULL_COND) || L.save || (B & 1) || !use_cs())
      return LIGDIFF_ERR_ARG;
    pemb_mols = B / 2;
  }
  if (flags & LIGDIFF_FWD_PROPERTIES) {
    if (flags & LIGDIFF_FWD_NULL_COND) {
      pemb = w[W_NULL_EMB]; pstride = 0;
    } else {
      if (properties == nullptr) return LIGDIFF_ERR_ARG;
      PilotEncWeights pw;
      for (int i = 0; i < 8; ++i) {
        pw.w0[i] = w[W_PENC(i, 0)]; pw.b0[i] = w[W_PENC(i, 1)];
        pw.W2[i] = w[W_PENC(i, 2)]; pw.b2[i] = w[W_PENC(i, 3)];
      }
      hipLaunchKernelGGL(k_pilot_encode, dim3(pemb_mols), dim3(256), 0, s, properties, pw, F(L.pstack), S(L.pen_pre));
      gemm_bias_act<ACT_GELU, false>(F(L.pstack), 2048, pemb_mols, 2048, w[W_PAR0_W], 2048, w[W_PAR0_B], 1, F(L.pg), 256,
                                     nullptr, s, S(L.pg_pre));
      hipLaunchKernelGGL(k_pilot_combine, dim3(pemb_mols), dim3(256), 0, s, F(L.pg), w[W_PAR2_W], w[W_PAR2_B], F(L.pstack), F(L.pemb),
                         S(L.pz), S(L.pw));
      pemb = F(L.pemb); pstride = 256;
    }
  }

  // --- input embeddings (model1.py:709-711)
  {
    GemmCore gc = core_blocks(N, LD_A, 256, w[W_ATOM_W], LD_A, 1, 256);
    ALoadScalar al{x, LD_A, N, LD_A};
    EpiEmbed e;
    e.M = N; e.ba = w[W_ATOM_B]; e.ga = w[W_ATOM_G]; e.bea = w[W_ATOM_BE];
    e.pos = pos; e.Wp = w[W_POS_W]; e.bp = w[W_POS_B]; e.gp = w[W_POS_G]; e.bep = w[W_POS_BE];
    e.temb = F(L.temb); e.temb_stride = tstride; e.pemb = pemb; e.pemb_stride = pstride;
    e.atom_mol = g->atom_mol; e.H = F(L.h0); e.ua_out = S(L.ua);
    PB("embed");
    if (use_cs()) {
      // h = GELU(LN(x Wa^T + ba)) + R,  R = GELU(LN(pos Wp^T + bp)) + t_emb (+ prop_emb); the same launch
      // zero-extends x and Wa to 128 columns for the GEMM (64 extra workgroups for the weight rows)
      float* R = F(L.xn1[0]);   // free until the attention layers
      hipLaunchKernelGGL(k_pos_embed, dim3(ceil_div(N, 4) + 64), dim3(256), 0, s, pos, w[W_POS_W], w[W_POS_B], w[W_POS_G],
                         w[W_POS_BE], F(L.temb), tstride, pemb, pstride, g->atom_mol, R, N, pemb_mols, x, w[W_ATOM_W],
                         F(L.xpad), F(L.wpad));
      GemmCore gp = core_blocks(N, 128, 256, F(L.wpad), 128, 1, 256);
      EpiLN<ACT_NONE, ACT_GELU, true> el;
      memset(&el, 0, sizeof el);
      el.M = N; el.bias[0] = w[W_ATOM_B]; el.gamma[0] = w[W_ATOM_G]; el.beta[0] = w[W_ATOM_BE]; el.eps = 1e-5f;
      el.C = F(L.h0); el.ldc = 256; el.resid = R; el.ldr = 256; el.pre_out = S(L.ua);
      launch_gemm_cs(gp, APtrPlain{F(L.xpad), 128}, el, 1, s);
    } else {
      launch_gemm<8, W_SCALAR>(gc, al, e, 1, s);
    }
    PE();
  }

  // --- E(3) message-passing layers (model1.py:721-725, :86-114)
  const float* h_in = F(L.h0);
  // inference on the fp32 pipe: the three edge launches of a layer run as one fused kernel
  // (their vector loads and LDS-DMAs need 16-byte aligned weight tensors)
  bool w_aligned = true;
  for (int i = 0; i < W_CONS_0; ++i) w_aligned = w_aligned && (w[i] == nullptr || (reinterpret_cast<uintptr_t>(w[i]) & 15) == 0);
  const FwdPlan plan = forward_plan(N, E, B, g->max_mol_atoms);
  const bool edge_chain = w_aligned && plan.edge_chain && !L.save && g_split == nullptr && E > 0 && (int64_t)N * 512 < (1LL << 31);
  // graphs of high in-degree (fully connected ligands): the fused edge kernel sums each destination's messages per
  // 32-edge slab itself and the [E,256] message tensor never reaches memory; the partial sums take the message
  // buffer's place (at most N + E / 32 + B + 3 rows, so the form needs E >= that)
  const bool edge_reduce = edge_chain && plan.edge_reduce && (int64_t)N + L.n_slab_cap <= (int64_t)E;
  // likewise the node update of a layer together with what follows it (the next layer's per-node edge
  // projections, or the first attention layer's pre-norm and in-projection), and everything between two
  // attention kernels (out-projection, feed-forward block, the next layer's pre-norm and in-projection)
  const bool node_chain = w_aligned && plan.node_chain && !L.save && g_split == nullptr && N > 0;
  // (the node and attention-layer kernels stage ONE adaLN row: calls whose molecules share the timestep)
  const bool dit_chain = node_chain && drop == nullptr && mstride == 0;
  // one molecule per wave, attention inside the attention-layer kernel (no attention launches, no qkv in memory)
  const bool mol_wave = dit_chain && plan.mol_wave && g->max_mol_atoms <= 32;
  int32_t* const mol_e0 = reinterpret_cast<int32_t*>(ws + L.mol_e0);
  int32_t* const mol_slab0 = reinterpret_cast<int32_t*>(ws + L.mol_slab0);
  int32_t* const slab_e0 = reinterpret_cast<int32_t*>(ws + L.slab_e0);
  int32_t* const slab_cnt = reinterpret_cast<int32_t*>(ws + L.slab_cnt);
  if (edge_reduce) {   // molecule-aligned 32-edge slabs, the same for every layer
    const int cap = (int)L.n_slab_cap;
    hipLaunchKernelGGL(k_slab_scan, dim3(1), dim3(1024), 0, s, g->mol_ptr, g->dst_ptr, B, mol_e0, mol_slab0);
    hipLaunchKernelGGL(k_slab_fill, dim3(ceil_div(B, 4) + ceil_div(cap, 256)), dim3(256), 0, s, mol_e0, mol_slab0, B, cap,
                       slab_e0, slab_cnt);
  }
  for (int l = 0; l < NL; ++l) {
    const float* pos_l = (l == 0) ? pos : F(L.pos_cur);
    // per-node halves of the first edge linear: P = h [W1a | W1b]^T
    if (!node_chain || l == 0) {
      GemmCore gc;
      memset(&gc, 0, sizeof gc);
      gc.M = N; gc.K = 256; gc.n_valid = 128; gc.ldw = 544;
      const float* W1 = w[W_E3(l, E3_EN0_W)];
      gc.W[0] = W1; gc.W[1] = W1 + (size_t)128 * 544; gc.W[2] = W1 + 256; gc.W[3] = W1 + (size_t)128 * 544 + 256;
      EpiBiasAct<4, ACT_NONE, false> e;
      memset(&e, 0, sizeof e);
      e.M = N; e.n_valid = 128; e.col_block = 128; e.C = F(L.P[l]); e.ldc = 512; e.alpha = 1.0f;
      PB("e3_%d.node_proj", l);
      if (use_cs()) {
        GemmCore g2;
        memset(&g2, 0, sizeof g2);
        g2.M = N; g2.K = 256; g2.n_valid = 256; g2.ldw = 544; g2.W[0] = W1; g2.W[1] = W1 + 256;
        EpiBiasAct<8, ACT_NONE, false> e2;
        memset(&e2, 0, sizeof e2);
        e2.M = N; e2.n_valid = 256; e2.col_block = 256; e2.C = F(L.P[l]); e2.ldc = 512; e2.alpha = 1.0f;
        if (edge_chain) e2.bias[1] = w[W_E3(l, E3_EN0_B)];   // the fused edge kernel expects b1 inside Pb
        launch_gemm_cs(g2, APtrPlain{h_in, 256}, e2, 2, s);
      } else {
        launch_gemm_plain<4>(gc, h_in, 256, e, 4, s);
      }
      PE();
    }
    if (E > 0) {
      // the fused edge kernel computes the geometry and the radial MLP itself (LIGDIFF_EDGE_GEOM=0: A/B switch)
      static const bool geom_fused = [] { const char* e = getenv("LIGDIFF_EDGE_GEOM"); return !(e != nullptr && e[0] == '0'); }();
      const bool edge_geom_inside = edge_chain && geom_fused;
      if (!edge_geom_inside) {
        PB("e3_%d.edge_geom", l);
        hipLaunchKernelGGL(k_edge_geom, dim3(ceil_div(E, 128)), dim3(256), 0, s, pos_l, g->src, g->dst,
                           w[W_E3(l, E3_RB0_W)], w[W_E3(l, E3_RB0_B)], w[W_E3(l, E3_RB2_W)], w[W_E3(l, E3_RB2_B)],
                           F(L.rel[l]), F(L.rbf[l]), S(L.dist[l]), S(L.r1pre[l]), E);
        PE();
      }
      if (edge_chain) {
        EdgeChainArgs ea;
        memset(&ea, 0, sizeof ea);
        ea.E = E; ea.rbf = F(L.rbf[l]); ea.rel = F(L.rel[l]); ea.src = g->src; ea.dst = g->dst; ea.P = F(L.P[l]);
        ea.Wc = w[W_E3(l, E3_EN0_W)] + 512; ea.ldwc = 544;
        ea.W2 = w[W_E3(l, E3_EN2_W)]; ea.b2 = w[W_E3(l, E3_EN2_B)]; ea.gamma = w[W_E3(l, E3_EN3_G)]; ea.beta = w[W_E3(l, E3_EN3_B)];
        ea.Wc1 = w[W_E3(l, E3_CN0_W)]; ea.bc1 = w[W_E3(l, E3_CN0_B)]; ea.wc2 = w[W_E3(l, E3_CN2_W)]; ea.bc2 = w[W_E3(l, E3_CN2_B)];
        ea.msg = F(L.msg[l]); ea.wrel = F(L.wrel[l]);
        if (edge_reduce) { ea.part = F(L.msg[l]); ea.slab_e0 = slab_e0; ea.slab_cnt = slab_cnt; ea.n_slabs = (int)L.n_slab_cap; }
        if (edge_geom_inside) {
          ea.pos = pos_l; ea.rw0 = w[W_E3(l, E3_RB0_W)]; ea.rb0 = w[W_E3(l, E3_RB0_B)];
          ea.rW2 = w[W_E3(l, E3_RB2_W)]; ea.rb2 = w[W_E3(l, E3_RB2_B)];
        }
        PB("e3_%d.edge_chain", l);
        launch_edge_chain(ea, s);
        PE();
      } else {
      {  // a1 = SiLU(Pa[src] + Pb[dst] + Wc rbf + b1)
        GemmCore gc;
        memset(&gc, 0, sizeof gc);
        gc.M = E; gc.K = 32; gc.n_valid = 256; gc.ldw = 544; gc.W[0] = w[W_E3(l, E3_EN0_W)] + 512;
        EpiEdgePre e{E, w[W_E3(l, E3_EN0_B)], F(L.P[l]), g->src, g->dst, F(L.a1[l]), S(L.pre_e[l])};
        PB("e3_%d.edge_pre", l);
        launch_gemm_plain<8>(gc, F(L.rbf[l]), 32, e, 1, s);
        PE();
      }
      {  // msg = LN(a1 W2^T + b2)
        GemmCore gc = core_blocks(E, 256, 256, w[W_E3(l, E3_EN2_W)], 256, 1, 256);
        EpiLN<ACT_NONE, ACT_NONE, false> e;
        memset(&e, 0, sizeof e);
        e.M = E; e.bias[0] = w[W_E3(l, E3_EN2_B)]; e.gamma[0] = w[W_E3(l, E3_EN3_G)]; e.beta[0] = w[W_E3(l, E3_EN3_B)];
        e.eps = 1e-5f; e.C = F(L.msg[l]); e.ldc = 256; e.pre_out = S(L.u_m[l]);
        PB("e3_%d.edge_mlp2", l);
        launch_gemm_plain<8>(gc, F(L.a1[l]), 256, e, 1, s);
        PE();
      }
      {  // wrel = rel * tanh(w2 . SiLU(msg Wc1^T + b) + b2)
        GemmCore gc = core_blocks(E, 256, 128, w[W_E3(l, E3_CN0_W)], 256, 1, 128);
        EpiCoord e{E, w[W_E3(l, E3_CN0_B)], w[W_E3(l, E3_CN2_W)], w[W_E3(l, E3_CN2_B)],
                          F(L.rel[l]), F(L.wrel[l]), S(L.hc_pre[l]), S(L.cw[l])};
        PB("e3_%d.coord", l);
        // E = 2 x atoms rows already give two 4-wave workgroups per CU; the eight-wave form measured
        // slower here (its dot-product head needs the mailbox exchange and a barrier)
        if (use_gemm2()) {
          launch_gemm_rows128(gc, APtrPlain{F(L.msg[l]), 256}, e, 1, s);
        } else {
          ALoadPlain alc{F(L.msg[l]), 256, E, 256};
          launch_gemm<4, W_ALIGNED>(gc, alc, e, 1, s);
        }
        PE();
      }
      }
    }
    PB("e3_%d.aggregate", l);
    if (edge_reduce)
      hipLaunchKernelGGL(k_aggregate<true>, dim3(ceil_div(N, 8)), dim3(256), 0, s, F(L.msg[l]), F(L.wrel[l]), g->dst_ptr,
                         g->src_ptr, g->src_slot, pos, F(L.pos_upd), F(L.pos_cur), F(L.agg[l]), (float*)nullptr, N,
                         l == 0 ? 1 : 0, g->atom_mol, mol_e0, mol_slab0);
    else
      hipLaunchKernelGGL(k_aggregate<false>, dim3(ceil_div(N, 8)), dim3(256), 0, s, F(L.msg[l]), F(L.wrel[l]), g->dst_ptr,
                         g->src_ptr, g->src_slot, pos, F(L.pos_upd), F(L.pos_cur), F(L.agg[l]), (float*)nullptr, N,
                         l == 0 ? 1 : 0);
    PE();
    if (node_chain) {
      NodeChainArgs na;
      memset(&na, 0, sizeof na);
      na.M = N; na.h = h_in; na.agg = F(L.agg[l]); na.hout = F(L.h_e3[l]);
      na.Wn1 = w[W_E3(l, E3_NN0_W)]; na.b1 = w[W_E3(l, E3_NN0_B)]; na.Wn2 = w[W_E3(l, E3_NN2_W)]; na.b2 = w[W_E3(l, E3_NN2_B)];
      na.gamma = w[W_E3(l, E3_NN3_G)]; na.beta = w[W_E3(l, E3_NN3_B)];
      PB("e3_%d.node_chain", l);
      if (l < NL - 1) {
        na.Wt = w[W_E3(l + 1, E3_EN0_W)]; na.bt = edge_chain ? w[W_E3(l + 1, E3_EN0_B)] : nullptr; na.tout = F(L.P[l + 1]);
        launch_node_chain(na, NODE_TAIL_PROJ, s);
      } else if (dit_chain && !mol_wave) {
        na.Wt = w[W_DIT(0, DIT_INP_W)]; na.bt = w[W_DIT(0, DIT_INP_B)]; na.tout = F(L.qkv[0]); na.mod = F(L.mod[0]);
        launch_node_chain(na, NODE_TAIL_QKV, s);
      } else {
        launch_node_chain(na, NODE_TAIL_NONE, s);
      }
      PE();
      h_in = F(L.h_e3[l]);
      continue;
    }
    {  // hid = SiLU([h | agg] Wn1^T + b)
      GemmCore gc = core_blocks(N, 512, 256, w[W_E3(l, E3_NN0_W)], 512, 1, 256);
      ALoadConcat al{h_in, F(L.agg[l]), 256, 256, 256, N, 512};
      APtrConcat ap{h_in, F(L.agg[l]), 256, 256, 256};
      EpiBiasAct<8, ACT_SILU, false> e;
      memset(&e, 0, sizeof e);
      e.M = N; e.n_valid = 256; e.col_block = 256; e.bias[0] = w[W_E3(l, E3_NN0_B)]; e.C = F(L.hid[l]); e.ldc = 256;
      e.alpha = 1.0f; e.pre_out = S(L.hid_pre[l]);
      PB("e3_%d.node_mlp1", l);
      if (use_cs())
        launch_gemm_cs(gc, ap, e, 1, s);
      else if (use_gemm2())
        hipLaunchKernelGGL((gemm_rows2_kernel<8, APtrConcat, EpiBiasAct<8, ACT_SILU, false>>), dim3(ceil_div(N, G_BM), 1),
                           dim3(256), 0, s, gc, ap, e);
      else
        launch_gemm<8, W_ALIGNED>(gc, al, e, 1, s);
      PE();
    }
    {  // h = h + LN(hid Wn2^T + b)
      GemmCore gc = core_blocks(N, 256, 256, w[W_E3(l, E3_NN2_W)], 256, 1, 256);
      EpiLN<ACT_NONE, ACT_NONE, true> e;
      memset(&e, 0, sizeof e);
      e.M = N; e.bias[0] = w[W_E3(l, E3_NN2_B)]; e.gamma[0] = w[W_E3(l, E3_NN3_G)]; e.beta[0] = w[W_E3(l, E3_NN3_B)];
      e.eps = 1e-5f; e.C = F(L.h_e3[l]); e.ldc = 256; e.resid = h_in; e.ldr = 256; e.pre_out = S(L.u_h[l]);
      PB("e3_%d.node_mlp2", l);
      launch_gemm_plain<8>(gc, F(L.hid[l]), 256, e, 1, s);
      PE();
    }
    h_in = F(L.h_e3[l]);
  }

  // --- adaLN attention layers over each molecule's own atoms (model1.py:728-738, :279-302)
  const int max_n = g->max_mol_atoms;
  const float* x_in = h_in;
  if (!dit_chain) {
    PB("dit0.ln_mod");
    hipLaunchKernelGGL(k_ln_mod, dim3(ceil_div(N, 16)), dim3(256), 0, s, x_in, F(L.mod[0]), mstride, 0, 256,
                       g->atom_mol, F(L.xn1[0]), N);
    PE();
  }
  for (int l = 0; l < NL; ++l) {
    if (!dit_chain) {
      PB("dit%d.qkv", l);
      gemm_bias_act<ACT_NONE, false>(F(L.xn1[l]), 256, N, 256, w[W_DIT(l, DIT_INP_W)], 256, w[W_DIT(l, DIT_INP_B)], 3,
                                     F(L.qkv[l]), 768, nullptr, s);
      PE();
    }
    if (!mol_wave) {
    PB("dit%d.attention", l);
    {
      const int tk = (max_n + 31) / 32;
      const size_t lds = (size_t)4 * 2 * (tk * 32) * 33 * 4;
      const int blocks = ceil_div((long)B * 8 * tk, 4);
      static const bool att_lds = getenv("LIGDIFF_ATT_LDS") != nullptr;   // A/B switch: LDS-staged form
      const float* qkv_l = F(L.qkv[l]);
      float* att_l = F(L.att[l]);
      float* lse_l = S(L.lse[l]);
      DropSite da;
      if (!make_drop_site(drop, l, LIGDIFF_DROP_ATTENTION, &da)) return LIGDIFF_ERR_ARG;
      if (da.thr != 0u) {   // train-mode dropout on the probabilities (always the register kernel)
        if (tk == 1)
          hipLaunchKernelGGL((k_attention_reg<1, true>), dim3(blocks), dim3(256), 0, s, qkv_l, g->mol_ptr, att_l, lse_l, B, da);
        else if (tk == 2)
          hipLaunchKernelGGL((k_attention_reg<2, true>), dim3(blocks), dim3(256), 0, s, qkv_l, g->mol_ptr, att_l, lse_l, B, da);
        else if (tk == 3)
          hipLaunchKernelGGL((k_attention_reg<3, true>), dim3(blocks), dim3(256), 0, s, qkv_l, g->mol_ptr, att_l, lse_l, B, da);
        else
          hipLaunchKernelGGL((k_attention_reg<4, true>), dim3(blocks), dim3(256), 0, s, qkv_l, g->mol_ptr, att_l, lse_l, B, da);
      } else if (!att_lds && tk == 1)
        hipLaunchKernelGGL(k_attention_reg<1>, dim3(blocks), dim3(256), 0, s, qkv_l, g->mol_ptr, att_l, lse_l, B, da);
      else if (!att_lds && tk == 2)
        hipLaunchKernelGGL(k_attention_reg<2>, dim3(blocks), dim3(256), 0, s, qkv_l, g->mol_ptr, att_l, lse_l, B, da);
      else if (!att_lds && tk == 3)
        hipLaunchKernelGGL(k_attention_reg<3>, dim3(blocks), dim3(256), 0, s, qkv_l, g->mol_ptr, att_l, lse_l, B, da);
      else if (!att_lds)
        hipLaunchKernelGGL(k_attention_reg<4>, dim3(blocks), dim3(256), 0, s, qkv_l, g->mol_ptr, att_l, lse_l, B, da);
      else if (tk == 1)
        hipLaunchKernelGGL(k_attention_mfma<1>, dim3(blocks), dim3(256), lds, s, F(L.qkv[l]), g->mol_ptr, F(L.att[l]), S(L.lse[l]), B);
      else if (tk == 2)
        hipLaunchKernelGGL(k_attention_mfma<2>, dim3(blocks), dim3(256), lds, s, F(L.qkv[l]), g->mol_ptr, F(L.att[l]), S(L.lse[l]), B);
      else if (tk == 3)
        hipLaunchKernelGGL(k_attention_mfma<3>, dim3(blocks), dim3(256), lds, s, F(L.qkv[l]), g->mol_ptr, F(L.att[l]), S(L.lse[l]), B);
      else
        hipLaunchKernelGGL(k_attention_mfma<4>, dim3(blocks), dim3(256), lds, s, F(L.qkv[l]), g->mol_ptr, F(L.att[l]), S(L.lse[l]), B);
    }
    PE();
    }
    if (dit_chain) {
      DitChainArgs da;
      memset(&da, 0, sizeof da);
      da.M = N; da.att = F(L.att[l]); da.x = x_in; da.xout = F(L.x2[l]);
      da.Wo = w[W_DIT(l, DIT_OUT_W)]; da.bo = w[W_DIT(l, DIT_OUT_B)];
      da.W1 = w[W_DIT(l, DIT_FF0_W)]; da.b1 = w[W_DIT(l, DIT_FF0_B)]; da.W2 = w[W_DIT(l, DIT_FF3_W)]; da.b2 = w[W_DIT(l, DIT_FF3_B)];
      da.mod = F(L.mod[l]);
      if (l < NL - 1) {
        da.mod_next = F(L.mod[l + 1]); da.Wqkv = w[W_DIT(l + 1, DIT_INP_W)]; da.bqkv = w[W_DIT(l + 1, DIT_INP_B)];
        da.qkv = F(L.qkv[l + 1]);
      }
      if (mol_wave) {
        da.Wqkv_self = w[W_DIT(l, DIT_INP_W)]; da.bqkv_self = w[W_DIT(l, DIT_INP_B)]; da.mol_ptr = g->mol_ptr; da.B = B;
        da.Wqkv = nullptr; da.bqkv = nullptr; da.qkv = nullptr; da.mod_next = nullptr;
      }
      if (l == NL - 1) {   // the last layer carries the atom and coordinate heads (model1.py:741-742)
        da.W0a = w[W_HEAD(0, HEAD_W0)]; da.b0a = w[W_HEAD(0, HEAD_B0)]; da.ga = w[W_HEAD(0, HEAD_G)]; da.bea = w[W_HEAD(0, HEAD_BE)];
        da.W3a = w[W_HEAD(0, HEAD_W3)]; da.b3a = w[W_HEAD(0, HEAD_B3)];
        da.W0p = w[W_HEAD(1, HEAD_W0)]; da.b0p = w[W_HEAD(1, HEAD_B0)]; da.gp = w[W_HEAD(1, HEAD_G)]; da.bep = w[W_HEAD(1, HEAD_BE)];
        da.W3p = w[W_HEAD(1, HEAD_W3)]; da.b3p = w[W_HEAD(1, HEAD_B3)];
        da.pos_upd = F(L.pos_upd); da.atom_pred = atom_pred; da.pos_pred = pos_pred;
      }
      PB("dit%d.chain", l);
      launch_dit_chain(da, l == NL - 1 ? DIT_TAIL_HEADS : mol_wave ? DIT_TAIL_NONE : DIT_TAIL_QKV, mol_wave, s);
      PE();
      x_in = F(L.x2[l]);
      continue;
    }
    {  // x1 = x + gate_msa * (att Wo^T + b);  xn2 = LN(x1)(1+scale_mlp)+shift_mlp
      GemmCore gc = core_blocks(N, 256, 256, w[W_DIT(l, DIT_OUT_W)], 256, 1, 256);
      EpiGate<true> e{N, w[W_DIT(l, DIT_OUT_B)], x_in, F(L.mod[l]), mstride, 512, F(L.mod[l]), 768, 1024,
                      g->atom_mol, F(L.x1[l]), F(L.xn2[l]), S(L.oproj[l])};
      PB("dit%d.out_proj", l);
      launch_gemm_plain<8>(gc, F(L.att[l]), 256, e, 1, s);
      PE();
    }
    DropSite dh, dout;
    if (!make_drop_site(drop, l, LIGDIFF_DROP_FFN_HIDDEN, &dh) || !make_drop_site(drop, l, LIGDIFF_DROP_FFN_OUT, &dout))
      return LIGDIFF_ERR_ARG;
    if ((dh.thr != 0u || dout.thr != 0u) && !use_cs()) return LIGDIFF_ERR_ARG;
    PB("dit%d.ffn_up", l);
    if (dh.thr != 0u) {   // ffh = Dropout(GELU(xn2 W0^T + b)), model1.py:268-270
      GemmCore gc = core_blocks(N, 256, 256, w[W_DIT(l, DIT_FF0_W)], 256, 4, 256);
      EpiBiasAct<8, ACT_GELU, false, true> e;
      memset(&e, 0, sizeof e);
      e.M = N; e.n_valid = 256; e.col_block = 256; e.C = F(L.ffh[l]); e.ldc = 1024; e.alpha = 1.0f;
      e.pre_out = S(L.ffh_pre[l]); e.drop = dh;
      for (int i = 0; i < 4; ++i) e.bias[i] = w[W_DIT(l, DIT_FF0_B)] + (size_t)i * 256;
      launch_gemm_cs(gc, APtrPlain{F(L.xn2[l]), 256}, e, 4, s);
    } else {
      gemm_bias_act<ACT_GELU, false>(F(L.xn2[l]), 256, N, 256, w[W_DIT(l, DIT_FF0_W)], 256, w[W_DIT(l, DIT_FF0_B)], 4,
                                     F(L.ffh[l]), 1024, nullptr, s, S(L.ffh_pre[l]));
    }
    PE();
    {  // x2 = x1 + gate_mlp * (ffh W3^T + b); next layer's xn1 fused
      GemmCore gc = core_blocks(N, 1024, 256, w[W_DIT(l, DIT_FF3_W)], 1024, 1, 256);
      if (l < NL - 1) {
        PB("dit%d.ffn_down", l);
        if (dout.thr != 0u) {   // Dropout on the FFN output before the gated residual, model1.py:271-272
          EpiGate<true, true> e{N, w[W_DIT(l, DIT_FF3_B)], F(L.x1[l]), F(L.mod[l]), mstride, 1280, F(L.mod[l + 1]), 0, 256,
                                g->atom_mol, F(L.x2[l]), F(L.xn1[l + 1]), S(L.fproj[l]), dout};
          launch_gemm_cs(gc, APtrPlain{F(L.ffh[l]), 1024}, e, 1, s);
        } else {
          EpiGate<true> e{N, w[W_DIT(l, DIT_FF3_B)], F(L.x1[l]), F(L.mod[l]), mstride, 1280, F(L.mod[l + 1]), 0, 256,
                          g->atom_mol, F(L.x2[l]), F(L.xn1[l + 1]), S(L.fproj[l])};
          launch_gemm_plain<8>(gc, F(L.ffh[l]), 1024, e, 1, s);
        }
        PE();
      } else {
        PB("dit%d.ffn_down", l);
        if (dout.thr != 0u) {
          EpiGate<false, true> e{N, w[W_DIT(l, DIT_FF3_B)], F(L.x1[l]), F(L.mod[l]), mstride, 1280, F(L.mod[l]), 0, 256,
                                 g->atom_mol, F(L.x2[l]), nullptr, S(L.fproj[l]), dout};
          launch_gemm_cs(gc, APtrPlain{F(L.ffh[l]), 1024}, e, 1, s);
        } else {
          EpiGate<false> e{N, w[W_DIT(l, DIT_FF3_B)], F(L.x1[l]), F(L.mod[l]), mstride, 1280, F(L.mod[l]), 0, 256,
                           g->atom_mol, F(L.x2[l]), nullptr, S(L.fproj[l])};
          launch_gemm_plain<8>(gc, F(L.ffh[l]), 1024, e, 1, s);
        }
        PE();
      }
    }
    x_in = F(L.x2[l]);
  }

  // --- heads (model1.py:741-746)
  if (!dit_chain) {
  {
    GemmCore gc;
    memset(&gc, 0, sizeof gc);
    gc.M = N; gc.K = 256; gc.n_valid = 256; gc.ldw = 256;
    gc.W[0] = w[W_HEAD(0, HEAD_W0)]; gc.W[1] = w[W_HEAD(1, HEAD_W0)];
    EpiLN<ACT_GELU, ACT_NONE, false> e;
    memset(&e, 0, sizeof e);
    e.M = N; e.eps = 1e-5f; e.C = F(L.yh); e.ldc = 512; e.pre_out = S(L.u_head);
    for (int hd = 0; hd < 2; ++hd) {
      e.bias[hd] = w[W_HEAD(hd, HEAD_B0)]; e.gamma[hd] = w[W_HEAD(hd, HEAD_G)]; e.beta[hd] = w[W_HEAD(hd, HEAD_BE)];
    }
    PB("head.hidden");
    launch_gemm_plain<8>(gc, x_in, 256, e, 2, s);
    PE();
  }
  {  // atom_pred = y_atom W3^T + b   (119 outputs)
    GemmCore gc = core_blocks(N, 256, LD_A, w[W_HEAD(0, HEAD_W3)], 256, 1, 128);
    EpiBiasAct<4, ACT_NONE, false> e;
    memset(&e, 0, sizeof e);
    e.M = N; e.n_valid = LD_A; e.col_block = 0; e.bias[0] = w[W_HEAD(0, HEAD_B3)]; e.C = atom_pred; e.ldc = LD_A;
    e.alpha = 1.0f;
    PB("head.atom");
    launch_gemm_plain<4>(gc, F(L.yh), 512, e, 1, s);
    PE();
  }
  PB("head.pos");
  hipLaunchKernelGGL(k_head_small<3>, dim3(ceil_div(N, 16)), dim3(256), 0, s, F(L.yh), 512, 256, w[W_HEAD(1, HEAD_W3)],
                     w[W_HEAD(1, HEAD_B3)], F(L.pos_upd), pos_pred, N);
  PE();
  }
  if (!(flags & LIGDIFF_FWD_SKIP_PROP_HEAD)) {
    if (prop_pred == nullptr) return LIGDIFF_ERR_ARG;
    hipLaunchKernelGGL(k_mean_pool, dim3(ceil_div(B, 4)), dim3(256), 0, s, x_in, g->mol_ptr, F(L.pooled), B);
    GemmCore gc = core_blocks(B, 256, 256, w[W_HEAD(2, HEAD_W0)], 256, 1, 256);
    EpiLN<ACT_GELU, ACT_NONE, false> e;
    memset(&e, 0, sizeof e);
    e.M = B; e.eps = 1e-5f; e.C = F(L.ph); e.ldc = 256; e.pre_out = S(L.u_ph);
    e.bias[0] = w[W_HEAD(2, HEAD_B0)]; e.gamma[0] = w[W_HEAD(2, HEAD_G)]; e.beta[0] = w[W_HEAD(2, HEAD_BE)];
    launch_gemm_plain<8>(gc, F(L.pooled), 256, e, 1, s);
    hipLaunchKernelGGL(k_head_small<15>, dim3(ceil_div(B, 16)), dim3(256), 0, s, F(L.ph), 256, 0, w[W_HEAD(2, HEAD_W3)],
                       w[W_HEAD(2, HEAD_B3)], (const float*)nullptr, prop_pred, B);
  }
  LD_CHECK_LAUNCH();
  return LIGDIFF_OK;
}
