// Internal interface between the translation units of the library.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "../../include/ligdiff.h"
#include "split_cache.h"
#include "weights.h"

// Byte offsets into the forward workspace (all 256-B aligned).
struct FwdLayout {
  int nl;                           // E(3) layers = attention layers of the call (ligdiff_options.layer_pairs)
  int64_t Bt;                       // rows of the time tensors: 1 when t is broadcast, else B
  size_t tf, t1, temb, stemb;       // Fourier features, time-MLP hidden, t_emb, SiLU(t_emb)
  size_t mod[LD_MAXL];                    // adaLN parameters [Bt,1536] per DiT layer
  size_t pemb, pstack, pg;          // PILOT embedding, stacked encoders, pareto hidden
  size_t h0;                        // [N,256] embedding output / residual stream
  size_t xpad, wpad;                // [N,128] / [256,128] zero-extended atom features and embedding weight
  size_t pos_upd, pos_cur;          // [N,3]
  size_t P[LD_MAXL], rel[LD_MAXL], rbf[LD_MAXL], a1[LD_MAXL], msg[LD_MAXL], wrel[LD_MAXL], agg[LD_MAXL], hid[LD_MAXL], h_e3[LD_MAXL];
  size_t xn1[LD_MAXL], qkv[LD_MAXL], att[LD_MAXL], x1[LD_MAXL], xn2[LD_MAXL], ffh[LD_MAXL], x2[LD_MAXL];
  size_t mol_e0, mol_slab0;         // [B+1] int32: first edge / first slab of each molecule (REDUCE form of the edge kernel)
  size_t slab_e0, slab_cnt;         // [n_slab_cap] int32: first edge / edge count of each slab
  int64_t n_slab_cap;               // slabs the tables hold: a multiple of 4 >= E / 32 + B
  size_t yh;                        // [N,512] head hidden (atom | pos)
  size_t pooled, ph;                // [B,256]
  // kept only with LIGDIFF_FWD_SAVE_FOR_BACKWARD (pre-activations the backward pass needs)
  int save;
  size_t t1_pre, ua, pen_pre, pg_pre, pz, pw;
  size_t pre_e[LD_MAXL], u_m[LD_MAXL], hc_pre[LD_MAXL], cw[LD_MAXL], dist[LD_MAXL], r1pre[LD_MAXL], hid_pre[LD_MAXL], u_h[LD_MAXL];
  size_t oproj[LD_MAXL], fproj[LD_MAXL], ffh_pre[LD_MAXL], lse[LD_MAXL], u_head, u_ph;
  size_t wsplit, wsplit_bytes;      // split-bf16 copies of the GEMM weights (split-operand mode only, else 0 bytes)
  size_t total;
};
// Per-call options (ligdiff_options): an *_opt entry point installs the caller's struct for the calling thread for
// the duration of the call; every choice below looks there first and at the process-wide defaults second.
struct CallOptions {
  const ligdiff_options* prev;
  explicit CallOptions(const ligdiff_options* o);
  ~CallOptions();
};
const ligdiff_options* call_options();
int check_options(const ligdiff_options* o);   // LIGDIFF_OK or LIGDIFF_ERR_ARG (a field out of range)
// Which forms the inference path takes for a launch set of these sizes: the call's options where they say, the
// fill / in-degree rules of launch.h where they do not.  "Wanted" only: run_forward still checks what each form
// needs (aligned weights, shared timestep, no saved activations, fp32 pipe, molecule sizes).
struct FwdPlan { bool edge_chain, node_chain, mol_wave, edge_reduce; };
FwdPlan forward_plan(int64_t n_atoms, int64_t n_edges, int64_t n_mols, int max_mol_atoms);
// True when guided sampling runs the conditional and unconditional batch as one doubled forward.
bool cfg_one_pass();
// Split-operand GEMM mode of the calling thread's current call (see ligdiff_set_split_gemm / ligdiff_options).
bool split_gemm_enabled();
// True unless an environment A/B switch routed the GEMMs to the register-staged kernels.
bool fwd_column_split_gemms();
void fwd_layout(FwdLayout* L, int64_t N, int64_t E, int32_t B, int flags);

// Measurement hook: the launch whose label equals `label` is bracketed by e0 / e1.
struct FwdProbe { const char* label; hipEvent_t e0, e1; int hits; };

// Library-internal flag (not part of the C ABI): the shared-timestep buffers of the workspace
// (temb [1,256] and the three adaLN rows mod[l] [1,1536]) were filled by the caller, so the forward
// skips its time path.  Only meaningful together with LIGDIFF_FWD_T_BROADCAST.
#define FWD_INTERNAL_TIME_GIVEN (1 << 20)
// Library-internal flag: the batch is a conditional half followed by an unconditional copy of the same
// molecules (classifier-free guidance in one pass): `properties` has n_mols / 2 rows and only the
// first n_mols / 2 molecules receive the property embedding.  Needs LIGDIFF_FWD_PROPERTIES, no null
// conditioning, no saved activations.
#define FWD_INTERNAL_HALF_COND (1 << 21)

// Bits a caller of the C ABI may set in `flags` (the LIGDIFF_FWD_* values); anything else is refused.
#define FWD_PUBLIC_FLAGS (LIGDIFF_FWD_PROPERTIES | LIGDIFF_FWD_NULL_COND | LIGDIFF_FWD_SKIP_PROP_HEAD | \
                          LIGDIFF_FWD_SAVE_FOR_BACKWARD | LIGDIFF_FWD_T_BROADCAST)
// Shape / pointer validation of a caller's ligdiff_graph, shared by every entry point that takes one
// (sizes positive and below 2^30, arrays present, largest molecule within LIGDIFF_MAX_ATOMS).
int check_graph(const ligdiff_graph* g);

// Time embedding + adaLN parameters for n_rows timesteps t[0..n_rows) with the kernels (and the
// per-row summation order) of the shared-timestep path: temb [n,256], mods[l] [n,1536] for l < n_layers;
// tf [n,256], t1 [n,512], stemb [n,256] are scratch.
int run_time_rows(const float* const* w, const float* t, int n_rows, float* tf, float* t1, float* temb,
                  float* stemb, float* const* mods, int n_layers, hipStream_t s);

int run_forward(const float* const* w, const ligdiff_graph* g, const float* x, const float* pos,
                const float* t, const int32_t* t_index_dev, const float* properties, int flags,
                float* atom_pred, float* pos_pred, float* prop_pred, char* ws, hipStream_t s,
                FwdProbe* probe = nullptr, SplitCache* split = nullptr, const ligdiff_dropout* drop = nullptr);

// Device-side description of one dropout site of a call (common.h: DropSite); thr == 0 when the
// call has no dropout or the site's p is 0.  Returns false for a probability outside [0, 1).
struct DropSite;
bool make_drop_site(const ligdiff_dropout* d, int layer, int kind, DropSite* out);
// True when the A/B environment switches left the default (column-split) GEMM kernels in place.
int check_dropout(const ligdiff_dropout* d);
