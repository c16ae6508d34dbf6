// extern "C" surface: argument validation and dispatch (see include/ligdiff.h).
#include "common.h"
#include "forward.h"
#include "weights.h"

extern "C" int ligdiff_version(void) { return LIGDIFF_VERSION; }

extern "C" const char* ligdiff_status_string(int status) {
  switch (status) {
    case LIGDIFF_OK: return "ok";
    case LIGDIFF_ERR_ARG: return "invalid argument (null pointer, negative size or unsupported shape)";
    case LIGDIFF_ERR_WORKSPACE: return "workspace too small";
    case LIGDIFF_ERR_LAUNCH: return "HIP launch or graph call failed";
    case LIGDIFF_ERR_BATCH: return "batch vector not sorted or edge index out of range";
    case LIGDIFF_ERR_TOO_MANY_ATOMS: return "a molecule has more atoms than the attention kernel supports";
    default: return "unknown status";
  }
}

extern "C" int ligdiff_weight_count(void) { return W_COUNT; }
extern "C" const char* ligdiff_weight_name(int slot) {
  return (slot >= 0 && slot < W_COUNT) ? weight_table()[slot].name : nullptr;
}
extern "C" int64_t ligdiff_weight_numel(int slot) {
  return (slot >= 0 && slot < W_COUNT) ? weight_table()[slot].numel : -1;
}

extern "C" size_t ligdiff_forward_workspace_bytes(int64_t n_atoms, int64_t n_edges, int32_t n_mols, int flags) {
  FwdLayout L;
  fwd_layout(&L, n_atoms, n_edges, n_mols, flags);
  return L.total;
}

int check_graph(const ligdiff_graph* g) {
  if (!g || g->n_atoms <= 0 || g->n_edges < 0 || g->n_mols <= 0) return LIGDIFF_ERR_ARG;
  if (g->n_atoms >= (1LL << 30) || g->n_edges >= (1LL << 30)) return LIGDIFF_ERR_ARG;
  if (!g->mol_ptr || !g->atom_mol || !g->dst_ptr || !g->src_ptr) return LIGDIFF_ERR_ARG;
  if (g->n_edges > 0 && (!g->src || !g->dst || !g->src_slot)) return LIGDIFF_ERR_ARG;
  if (g->max_mol_atoms <= 0) return LIGDIFF_ERR_ARG;
  if (g->max_mol_atoms > LIGDIFF_MAX_ATOMS) return LIGDIFF_ERR_TOO_MANY_ATOMS;
  return LIGDIFF_OK;
}

// Argument validation shared by ligdiff_forward and ligdiff_forward_timed.
static int check_forward_args(const float* const* weights, const ligdiff_graph* graph, const float* x, const float* pos,
                              const float* t, const int32_t* t_index_dev, int flags, const float* atom_pred,
                              const float* pos_pred, const void* workspace, size_t workspace_bytes) {
  int rc = check_graph(graph);
  if (rc) return rc;
  if (flags & ~FWD_PUBLIC_FLAGS) return LIGDIFF_ERR_ARG;   // library-internal bits are not part of the ABI
  if (!weights || !x || !pos || (!t && !t_index_dev) || !atom_pred || !pos_pred || !workspace) return LIGDIFF_ERR_ARG;
  for (int i = 0; i < W_COUNT; ++i)   // slots of layers beyond the call's layer_pairs may be NULL
    if (weights[i] == nullptr && weight_slot_layer(i) < call_layer_pairs()) return LIGDIFF_ERR_ARG;
  if (t_index_dev && !(flags & LIGDIFF_FWD_T_BROADCAST)) return LIGDIFF_ERR_ARG;
  if (workspace_bytes < ligdiff_forward_workspace_bytes(graph->n_atoms, graph->n_edges, graph->n_mols, flags))
    return LIGDIFF_ERR_WORKSPACE;
  return LIGDIFF_OK;
}

extern "C" int ligdiff_forward_train(const float* const* weights, const ligdiff_graph* graph, const float* x,
                                     const float* pos, const float* t, const int32_t* t_index_dev,
                                     const float* properties, int flags, float* atom_pred, float* pos_pred,
                                     float* prop_pred, void* workspace, size_t workspace_bytes, void* stream,
                                     const ligdiff_dropout* dropout) {
  int rc = check_forward_args(weights, graph, x, pos, t, t_index_dev, flags, atom_pred, pos_pred, workspace,
                              workspace_bytes);
  if (rc) return rc;
  rc = check_dropout(dropout);
  if (rc) return rc;
  return run_forward(weights, graph, x, pos, t, t_index_dev, properties, flags, atom_pred, pos_pred, prop_pred,
                     (char*)workspace, (hipStream_t)stream, nullptr, nullptr, dropout);
}

extern "C" size_t ligdiff_forward_workspace_bytes_opt(int64_t n_atoms, int64_t n_edges, int32_t n_mols, int flags,
                                                      const ligdiff_options* options) {
  if (check_options(options)) return 0;
  CallOptions scope(options);
  return ligdiff_forward_workspace_bytes(n_atoms, n_edges, n_mols, flags);
}
extern "C" int ligdiff_forward_opt(const float* const* weights, const ligdiff_graph* graph, const float* x,
                                   const float* pos, const float* t, const int32_t* t_index_dev,
                                   const float* properties, int flags, float* atom_pred, float* pos_pred,
                                   float* prop_pred, void* workspace, size_t workspace_bytes, void* stream,
                                   const ligdiff_dropout* dropout, const ligdiff_options* options) {
  if (check_options(options)) return LIGDIFF_ERR_ARG;
  CallOptions scope(options);
  return ligdiff_forward_train(weights, graph, x, pos, t, t_index_dev, properties, flags, atom_pred, pos_pred, prop_pred,
                               workspace, workspace_bytes, stream, dropout);
}
extern "C" int ligdiff_forward_timed(const float* const* weights, const ligdiff_graph* graph, const float* x,
                                     const float* pos, const float* t, const int32_t* t_index_dev,
                                     const float* properties, int flags, float* atom_pred, float* pos_pred,
                                     float* prop_pred, void* workspace, size_t workspace_bytes, void* stream,
                                     const char* label, int32_t reps, float* mean_ms_host);
extern "C" int ligdiff_forward_timed_opt(const float* const* weights, const ligdiff_graph* graph, const float* x,
                                         const float* pos, const float* t, const int32_t* t_index_dev,
                                         const float* properties, int flags, float* atom_pred, float* pos_pred,
                                         float* prop_pred, void* workspace, size_t workspace_bytes, void* stream,
                                         const char* label, int32_t reps, float* mean_ms_host,
                                         const ligdiff_options* options) {
  if (check_options(options)) return LIGDIFF_ERR_ARG;
  CallOptions scope(options);
  return ligdiff_forward_timed(weights, graph, x, pos, t, t_index_dev, properties, flags, atom_pred, pos_pred, prop_pred,
                               workspace, workspace_bytes, stream, label, reps, mean_ms_host);
}

extern "C" int ligdiff_forward(const float* const* weights, const ligdiff_graph* graph, const float* x,
                               const float* pos, const float* t, const int32_t* t_index_dev,
                               const float* properties, int flags, float* atom_pred, float* pos_pred,
                               float* prop_pred, void* workspace, size_t workspace_bytes, void* stream) {
  return ligdiff_forward_train(weights, graph, x, pos, t, t_index_dev, properties, flags, atom_pred, pos_pred, prop_pred,
                               workspace, workspace_bytes, stream, nullptr);
}

// Measurement hook for bench.py: runs `reps` forwards on `stream` and returns the mean
// duration (ms, HIP events recorded on that stream) of the single launch labelled
// `label` ("dit1.ffn_up", "e3_1.aggregate", ...).  Synchronises the stream.
extern "C" int ligdiff_forward_timed(const float* const* weights, const ligdiff_graph* graph, const float* x,
                                     const float* pos, const float* t, const int32_t* t_index_dev,
                                     const float* properties, int flags, float* atom_pred, float* pos_pred,
                                     float* prop_pred, void* workspace, size_t workspace_bytes, void* stream,
                                     const char* label, int32_t reps, float* mean_ms_host) {
  int rc = check_forward_args(weights, graph, x, pos, t, t_index_dev, flags, atom_pred, pos_pred, workspace,
                              workspace_bytes);
  if (rc) return rc;
  if (!label || reps <= 0 || !mean_ms_host) return LIGDIFF_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  FwdProbe p;
  p.label = label; p.hits = 0;
  if (hipEventCreate(&p.e0) != hipSuccess || hipEventCreate(&p.e1) != hipSuccess) return LIGDIFF_ERR_LAUNCH;
  // cost of an empty event pair on this stream (the events bracket exactly one launch; what is
  // left after subtracting this is the kernel's own duration, as a kernel trace reports it)
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
  // split-operand mode: convert the weights in an unprobed pass so that no conversion launch falls
  // between the events
  SplitCache split;
  if (rc == 0 && split_gemm_enabled())
    rc = run_forward(weights, graph, x, pos, t, t_index_dev, properties, flags, atom_pred, pos_pred, prop_pred,
                     (char*)workspace, s, nullptr, &split);
  double total = 0.0;
  for (int i = 0; i < reps && rc == 0; ++i) {
    p.hits = 0;
    rc = run_forward(weights, graph, x, pos, t, t_index_dev, properties, flags, atom_pred, pos_pred, prop_pred,
                     (char*)workspace, s, &p, &split);
    if (rc == 0 && hipStreamSynchronize(s) != hipSuccess) rc = LIGDIFF_ERR_LAUNCH;
    if (rc == 0 && p.hits != 1) rc = LIGDIFF_ERR_ARG;  // unknown label
    float ms = 0.f;
    if (rc == 0 && hipEventElapsedTime(&ms, p.e0, p.e1) != hipSuccess) rc = LIGDIFF_ERR_LAUNCH;
    total += ms > empty_ms ? ms - empty_ms : 0.f;
  }
  (void)hipEventDestroy(p.e0);
  (void)hipEventDestroy(p.e1);
  *mean_ms_host = (float)(total / reps);
  return rc;
}

extern "C" int ligdiff_forward_state_offsets(int64_t n_atoms, int64_t n_edges, int32_t n_mols, int flags,
                                             size_t* hidden_offset, size_t* pos_offset) {
  if (n_atoms <= 0 || n_edges < 0 || n_mols <= 0 || !hidden_offset || !pos_offset) return LIGDIFF_ERR_ARG;
  FwdLayout L;
  fwd_layout(&L, n_atoms, n_edges, n_mols, flags);
  *hidden_offset = L.x2[L.nl - 1];
  *pos_offset = L.pos_cur;
  return LIGDIFF_OK;
}
extern "C" int ligdiff_forward_state_offsets_opt(int64_t n_atoms, int64_t n_edges, int32_t n_mols, int flags,
                                                 size_t* hidden_offset, size_t* pos_offset, const ligdiff_options* options) {
  if (check_options(options)) return LIGDIFF_ERR_ARG;
  CallOptions scope(options);
  return ligdiff_forward_state_offsets(n_atoms, n_edges, n_mols, flags, hidden_offset, pos_offset);
}
