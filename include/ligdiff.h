/*
 * ligdiff.h - C ABI of the MI355X (gfx950) ligand-diffusion hot path.
 *
 * Plain pointers and sizes only; every pointer named *_dev / every float* / int32_t* /
 * int64_t* argument is a DEVICE pointer unless its comment says "host".  The caller owns all
 * buffers and passes a workspace (sizes from the *_workspace_bytes functions); no entry point calls
 * hipMalloc or hipFree.  Every entry point takes the hipStream_t (as void*) it enqueues on and
 * returns an int status, 0 = ok.
 *
 * Per-call choices (GEMM mode, fused or separate kernels, concurrency hint, guidance in one or two passes)
 * travel in a ligdiff_options struct through the *_opt entry points; a call that is given options reads
 * nothing below for the fields it sets.  The process-wide setters are the DEFAULTS of calls without options
 * and are kept for existing callers (deprecated):
 *  - the GEMM mode (ligdiff_set_split_gemm, or LIGDIFF_GEMM_SPLIT in the environment): it changes what
 *    the *_workspace_bytes functions return, so size workspaces after choosing it and do not switch
 *    while another thread is inside a call that relies on the default;
 *  - the concurrency hint (ligdiff_set_concurrent_calls) and the chain-kernel choice (ligdiff_set_chain_mode);
 *  - LIGDIFF_CFG_TWO_PASS (environment, read at every sampling call and workspace query);
 *  - A/B switches read once from the environment (LIGDIFF_GEMM_V1, LIGDIFF_GEMM_NOCS, LIGDIFF_CHAIN / LIGDIFF_NO_CHAIN, LIGDIFF_CHAIN_MOL,
 *    LIGDIFF_ATT_LDS, LIGDIFF_GEMM_TILE64,
 *    LIGDIFF_ATT_BWD_V1, LIGDIFF_DW_BLOCKS);
 *  - the per-device scratch array of ligdiff_pack_csr_static.
 * HIP objects created and destroyed inside a call: ligdiff_ddpm_sample / ligdiff_ddim_sample with
 * use_graph != 0 capture, instantiate and destroy hipGraphs and synchronise the stream;
 * ligdiff_forward_timed / ligdiff_backward_timed create two events and synchronise the stream.  Everything else only
 * enqueues kernels.  Calls are re-entrant across devices and streams, the exceptions being the two
 * points above that say otherwise.
 *
 * The reference (Nishanth-nishu/diffusion_model_ligandmol_generation) is pure Python on
 * PyTorch with no FFI of its own; each entry point below names the reference code it
 * replaces.  The Python classes in diffusion_model_ligandmol_generation_amd/ keep the
 * reference's call signatures and reach these functions through ctypes.
 */
#ifndef LIGDIFF_H
#define LIGDIFF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: ligdiff_options gained layer_pairs, ligdiff_dropout holds LIGDIFF_MAX_LAYER_PAIRS probabilities per site kind,
 * the weights table has slots for LIGDIFF_MAX_LAYER_PAIRS layers of each kind */
#define LIGDIFF_VERSION 2

/* model constants the kernels are tiled for (model1.py:574-587 defaults) */
#define LIGDIFF_HIDDEN 256
#define LIGDIFF_HEADS 8
#define LIGDIFF_ATOM_DIM 119
#define LIGDIFF_PROP_DIM 15
#define LIGDIFF_RADIAL 32
#define LIGDIFF_E3_LAYERS 3
#define LIGDIFF_DIT_LAYERS 3
#define LIGDIFF_MAX_ATOMS 128
#define LIGDIFF_MAX_LAYER_PAIRS 8   /* layers of each kind the weights table has slots for */

enum ligdiff_status {
  LIGDIFF_OK = 0,
  LIGDIFF_ERR_ARG = 1,        /* null pointer / negative size / unsupported shape        */
  LIGDIFF_ERR_WORKSPACE = 2,  /* workspace smaller than ligdiff_*_workspace_bytes says   */
  LIGDIFF_ERR_LAUNCH = 3,     /* a HIP launch or graph call failed (hipGetLastError)     */
  LIGDIFF_ERR_BATCH = 4,      /* batch vector not sorted / edge index out of range       */
  LIGDIFF_ERR_TOO_MANY_ATOMS = 5
};

int ligdiff_version(void);
const char* ligdiff_status_string(int status);

/* ---------------------------------------------------------------------------------
 * Weights table.  One slot per tensor of the reference's state_dict that the hot path
 * reads (model1.py:595-655), fp32, [out, in] row-major exactly as nn.Linear stores it.
 * ligdiff_weight_name(i) returns the state_dict key of slot i so a host binding can fill
 * the table by name; ligdiff_weight_numel(i) the element count it must have.
 * The table has slots for LIGDIFF_MAX_LAYER_PAIRS E(3) layers ("equivariant_layers.<l>. ...") and as
 * many attention layers ("dit_layers.<l>. ..."): model1.py:627-634 builds num_layers // 2 of each.
 * A model with fewer leaves the slots of the layers it does not have NULL and passes its count in
 * ligdiff_options.layer_pairs (calls without options assume 3, i.e. num_layers = 6).
 * ------------------------------------------------------------------------------- */
int ligdiff_weight_count(void);
const char* ligdiff_weight_name(int slot);
int64_t ligdiff_weight_numel(int slot);

/* ---------------------------------------------------------------------------------
 * Packed ragged batch.  Replaces the PyG collate + per-call index arithmetic of
 * model1.py:88-112 / :728 (data_utils.py:898-914 defines the per-molecule arrays).
 * Built on device by ligdiff_pack_csr from the reference's own tensors.
 * ------------------------------------------------------------------------------- */
typedef struct ligdiff_graph {
  int64_t n_atoms;            /* N                                                      */
  int64_t n_edges;            /* E (directed)                                           */
  int32_t n_mols;             /* B                                                      */
  int32_t max_mol_atoms;      /* largest molecule, <= LIGDIFF_MAX_ATOMS                 */
  const int32_t* mol_ptr;     /* [B+1] first atom of each molecule                      */
  const int32_t* atom_mol;    /* [N]   molecule of each atom (int32 copy of `batch`)    */
  const int32_t* dst_ptr;     /* [N+1] CSR offsets, edges grouped by edge_index[1]      */
  const int32_t* src;         /* [E]   edge_index[0] in CSR order                       */
  const int32_t* dst;         /* [E]   edge_index[1] in CSR order                       */
  const int32_t* perm;        /* [E]   original edge id of each CSR slot (stable)       */
  const int32_t* src_ptr;     /* [N+1] offsets of the by-source list                    */
  const int32_t* src_slot;    /* [E]   CSR slots of the edges leaving each atom         */
} ligdiff_graph;

/* Bytes of int32 scratch ligdiff_pack_csr needs for N atoms / E edges. */
size_t ligdiff_pack_workspace_bytes(int64_t n_atoms, int64_t n_edges);

/*
 * edge_index [2,E] int64 and batch [N] int64 (sorted ascending) -> the int32 arrays of
 * ligdiff_graph (caller-allocated: mol_ptr B+1, atom_mol N, dst_ptr N+1, src/dst/perm E,
 * src_ptr N+1, src_slot E).  Edges keep their original order inside each destination
 * group, so sums over a group run in the order of a sequential index_add_.
 * status_dev[0] receives 0, or LIGDIFF_ERR_BATCH if validation failed; status_dev[1]
 * receives the largest molecule size.  Integer results are bit-exact and deterministic.
 */
int ligdiff_pack_csr(const int64_t* edge_index, int64_t n_edges, const int64_t* batch,
                     int64_t n_atoms, int32_t n_mols, int32_t* mol_ptr, int32_t* atom_mol,
                     int32_t* dst_ptr, int32_t* src, int32_t* dst, int32_t* perm,
                     int32_t* src_ptr, int32_t* src_slot, int32_t* status_dev,
                     void* workspace, size_t workspace_bytes, void* stream);

/*
 * ligdiff_pack_csr for callers that cannot allocate (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc): same inputs, outputs, status convention and bit-exact
 * results, but no workspace argument.  The counters live in a device array that is part of the
 * library's code object (4 MiB on every GPU the library runs on, held until the process ends), so
 * the call performs no hipMalloc / hipFree and needs none made for it.  n_atoms above
 * LIGDIFF_PACK_STATIC_MAX_ATOMS returns LIGDIFF_ERR_WORKSPACE (use ligdiff_pack_csr).  All calls on one
 * GPU share that GPU's array: calls that overlap in time on one GPU (two streams, two threads) race;
 * calls on different GPUs do not interact.  `stream` must belong to the CURRENT device (the array is resolved
 * for the current device; a stream of another device returns LIGDIFF_ERR_ARG).  The first call on a device
 * resolves the array's device address (hipGetSymbolAddress) and should be made outside stream capture.
 * No counterpart in the reference (it packs nothing; PyG collate + per-call index arithmetic,
 * model1.py:88-112, main.py:205-212).
 */
#define LIGDIFF_PACK_STATIC_MAX_ATOMS 262144
int ligdiff_pack_csr_static(const int64_t* edge_index, int64_t n_edges, const int64_t* batch,
                            int64_t n_atoms, int32_t n_mols, int32_t* mol_ptr, int32_t* atom_mol,
                            int32_t* dst_ptr, int32_t* src, int32_t* dst, int32_t* perm,
                            int32_t* src_ptr, int32_t* src_slot, int32_t* status_dev, void* stream);

/* ---------------------------------------------------------------------------------
 * Per-call options.  The inference path has two forms of several layers - fused "chain" kernels and
 * separate launches - that compute the same layers (model1.py:60-114, :255-302) and agree to fp32
 * rounding but not bit for bit, and by default picks per launch by how well it fills the chip.  A
 * caller that cuts ONE request into several calls (halves on two streams, shards on several GPUs) and
 * wants the pieces to produce the bits of the whole plans once with ligdiff_plan on the whole
 * request's sizes and passes the resolved struct to every piece.
 * Every field: -1 (or 0 for concurrent_calls) = not set, the process default applies.
 * No counterpart in the reference (one device, one kernel per ATen op).
 * ------------------------------------------------------------------------------- */
typedef struct ligdiff_options {
  int32_t gemm_mode;         /* 0 fp32 matrix pipe, 1 split-operand bf16 pipe (see ligdiff_set_split_gemm)   */
  int32_t chain_mode;        /* 0 never fused, 1 by fill, 2 always (see ligdiff_set_chain_mode); the four    */
                             /* fields below override it form by form                                        */
  int32_t concurrent_calls;  /* 1..16: calls sharing the GPU (see ligdiff_set_concurrent_calls); affects     */
                             /* only choices between bit-identical forms once the fields below are set       */
  int32_t edge_chain;        /* 0 / 1: fused edge path of the E(3) layers                                    */
  int32_t node_chain;        /* 0 / 1: fused node update and attention-layer kernels                         */
  int32_t mol_wave;          /* 0 / 1: molecule-per-wave attention layers (needs molecules of <= 32 atoms)   */
  int32_t edge_reduce;       /* 0 / 1: the fused edge path sums messages per destination itself              */
  int32_t cfg_two_pass;      /* 0 guided sampling as one doubled forward, 1 as two forwards                  */
  int32_t layer_pairs;       /* 1..LIGDIFF_MAX_LAYER_PAIRS: E(3) layers = attention layers = num_layers // 2 */
                             /* of the model whose weights table the call gets (model1.py:627-634); not set  */
                             /* = 3 (num_layers = 6).  The table's slots of further layers are ignored.      */
} ligdiff_options;

/* Sets every field of *o to "not set". */
void ligdiff_options_init(ligdiff_options* o);

/* Fills *o with what a call WITHOUT options would currently use for gemm_mode, chain_mode, concurrent_calls and
 * cfg_two_pass (the process-wide setters and the environment); the kernel-form fields stay unset.  Read-only. */
void ligdiff_default_options(ligdiff_options* o);

/* Resolves the kernel-form fields (edge_chain, node_chain, mol_wave, edge_reduce) that `in` (may be NULL)
 * leaves unset into 0 / 1 for a request of these sizes on the current device, as the library would choose
 * for ONE call over the whole request; the other fields are copied.  Pass *out to every call that
 * processes a part of the request: the parts then take the same forms, whatever their own sizes. */
int ligdiff_plan(int64_t n_atoms, int64_t n_edges, int32_t n_mols, int32_t max_mol_atoms,
                 const ligdiff_options* in, ligdiff_options* out);

/* ---------------------------------------------------------------------------------
 * Score-network forward.  Replaces ResearchAccurateDiffusionModel.forward in eval mode
 * (model1.py:699-755): embeddings, 3 E(3) message-passing layers (model1.py:86-114),
 * 3 adaLN attention layers over each molecule's own atoms (model1.py:279-302; equal to
 * the reference's pad-to-max_atoms + key_padding_mask for every real atom), heads.
 * ------------------------------------------------------------------------------- */
enum ligdiff_forward_flags {
  LIGDIFF_FWD_PROPERTIES = 1,   /* properties [B,15] given (model1.py:714-718)           */
  LIGDIFF_FWD_NULL_COND = 2,    /* classifier_free_guidance=True: learned null vector     */
  LIGDIFF_FWD_SKIP_PROP_HEAD = 4, /* do not compute prop_pred (samplers ignore it)        */
  LIGDIFF_FWD_SAVE_FOR_BACKWARD = 8, /* keep activations in the workspace for backward    */
  LIGDIFF_FWD_T_BROADCAST = 16  /* t has ONE element used for every molecule              */
};

size_t ligdiff_forward_workspace_bytes(int64_t n_atoms, int64_t n_edges, int32_t n_mols, int flags);

/*
 * weights: host array of ligdiff_weight_count() device pointers.
 * x [N,119], pos [N,3], t [B] (or [1] with LIGDIFF_FWD_T_BROADCAST; the value fed to the
 * Fourier embedding: integer step for the samplers, c_noise for the trainer),
 * properties [B,15] or NULL.  Outputs atom_pred [N,119], pos_pred [N,3], prop_pred [B,15]
 * (may be NULL with LIGDIFF_FWD_SKIP_PROP_HEAD).
 * t_index_dev: if non-NULL, a device int32; the time value used is (float)*t_index_dev
 * for every molecule and `t` is ignored (lets a captured graph step through time).
 */
int ligdiff_forward(const float* const* weights, const ligdiff_graph* graph, const float* x,
                    const float* pos, const float* t, const int32_t* t_index_dev,
                    const float* properties, int flags, float* atom_pred, float* pos_pred,
                    float* prop_pred, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------
 * Train-mode dropout of the attention layers (model1.py:263-274: nn.MultiheadAttention(dropout=0.1)
 * on the attention probabilities, nn.Dropout(0.1) after the FFN's GELU and after its second Linear).
 * A mask element is a pure function of (seed, site, global element index) through Philox4x32-10, so
 * forward and backward regenerate the same mask from the same struct and a batch sharded over GPUs
 * draws the mask of the unsharded batch (atom_offset = global index of this call's first atom).
 * Kept elements are scaled by 1 / (1 - p), as torch.nn.functional.dropout does; p = 0 disables a
 * site.  The random stream is not torch's: parity with the reference holds for a given mask, which
 * ligdiff_dropout_mask exports.
 * ------------------------------------------------------------------------------- */
typedef struct ligdiff_dropout {
  uint64_t seed;            /* Philox key of this forward / backward pair                         */
  int64_t atom_offset;      /* global index of atom 0 of this call                                */
  float p_attention[LIGDIFF_MAX_LAYER_PAIRS];   /* per attention layer: probabilities, FFN hidden, FFN output */
  float p_ffn_hidden[LIGDIFF_MAX_LAYER_PAIRS];  /* (entries of layers the model does not have are ignored)    */
  float p_ffn_out[LIGDIFF_MAX_LAYER_PAIRS];
  float p_consistency[2];   /* consistency module (model1.py:390, :401): Dropout inside the atom-type */
                            /* and the bond predictor; read by ligdiff_consistency_train only     */
} ligdiff_dropout;

enum ligdiff_dropout_kind {
  LIGDIFF_DROP_ATTENTION = 0, LIGDIFF_DROP_FFN_HIDDEN = 1, LIGDIFF_DROP_FFN_OUT = 2,
  LIGDIFF_DROP_CONS_ATOM = 3,   /* layer ignored; [n_atoms, 128], row = atom (global index)        */
  LIGDIFF_DROP_CONS_BOND = 4    /* layer ignored; [n_pairs, 128], row = ordered pair of THIS call   */
};

/* ligdiff_forward / ligdiff_backward with dropout (NULL = none: identical to the plain entry points).
 * Needs the default GEMM kernels (no LIGDIFF_GEMM_V1 / LIGDIFF_GEMM_NOCS in the environment). */
int ligdiff_forward_train(const float* const* weights, const ligdiff_graph* graph, const float* x,
                          const float* pos, const float* t, const int32_t* t_index_dev,
                          const float* properties, int flags, float* atom_pred, float* pos_pred,
                          float* prop_pred, void* workspace, size_t workspace_bytes, void* stream,
                          const ligdiff_dropout* dropout);

/* ligdiff_forward_train / ligdiff_forward_workspace_bytes under per-call options (NULL = the defaults). */
size_t ligdiff_forward_workspace_bytes_opt(int64_t n_atoms, int64_t n_edges, int32_t n_mols, int flags,
                                           const ligdiff_options* options);
int ligdiff_forward_opt(const float* const* weights, const ligdiff_graph* graph, const float* x,
                        const float* pos, const float* t, const int32_t* t_index_dev,
                        const float* properties, int flags, float* atom_pred, float* pos_pred,
                        float* prop_pred, void* workspace, size_t workspace_bytes, void* stream,
                        const ligdiff_dropout* dropout, const ligdiff_options* options);

/* Writes the mask scale (0 or 1 / (1 - p)) of every element of one dropout site for atoms
 * [0, n_atoms) of a call made with `dropout`: kind LIGDIFF_DROP_FFN_HIDDEN -> out [n_atoms, 1024],
 * LIGDIFF_DROP_FFN_OUT -> out [n_atoms, 256], LIGDIFF_DROP_ATTENTION -> out [n_atoms, 8, 128] indexed
 * (query atom, head, key index within the query's molecule); the two consistency kinds -> out
 * [n_atoms or n_pairs (passed as n_atoms), 128]. */
int ligdiff_dropout_mask(const ligdiff_dropout* dropout, int32_t layer, int32_t kind, int64_t n_atoms,
                         float* out, void* stream);

/* ---------------------------------------------------------------------------------
 * Score-network backward.  Replaces autograd's reverse pass through model1.py:699-755
 * (train.py:195).  `fwd_workspace` must be the workspace of a ligdiff_forward call made
 * with LIGDIFF_FWD_SAVE_FOR_BACKWARD (per-molecule t, same graph / x / pos / properties /
 * flags).  grads: host array in the weights-table order of device pointers to PRE-ZEROED
 * fp32 tensors shaped like the weights; parameter gradients are ACCUMULATED into them
 * (float atomics over row splits, so the last bits vary from run to run).  The entries of
 * "__time_freqs__" and of the moldiff_consistency.* slots may be NULL: no loss reaches them.
 * d_atom [N,119], d_pos [N,3], d_prop [B,15] (NULL = no gradient through prop_pred).
 * ------------------------------------------------------------------------------- */
size_t ligdiff_backward_workspace_bytes(int64_t n_atoms, int64_t n_edges, int32_t n_mols);
int ligdiff_backward(const float* const* weights, float* const* grads, const ligdiff_graph* graph,
                     const float* x, const float* pos, const float* properties, int flags,
                     const float* d_atom, const float* d_pos, const float* d_prop,
                     const void* fwd_workspace, void* workspace, size_t workspace_bytes, void* stream);

/* ligdiff_backward for a forward pass made by ligdiff_forward_train with the same `dropout`. */
int ligdiff_backward_train(const float* const* weights, float* const* grads, const ligdiff_graph* graph,
                           const float* x, const float* pos, const float* properties, int flags,
                           const float* d_atom, const float* d_pos, const float* d_prop,
                           const void* fwd_workspace, void* workspace, size_t workspace_bytes, void* stream,
                           const ligdiff_dropout* dropout);
/* The same under per-call options (only gemm_mode matters to the backward pass; the forward pass whose
 * workspace it reads must have run under the same gemm_mode). */
size_t ligdiff_backward_workspace_bytes_opt(int64_t n_atoms, int64_t n_edges, int32_t n_mols,
                                            const ligdiff_options* options);
int ligdiff_backward_opt(const float* const* weights, float* const* grads, const ligdiff_graph* graph,
                         const float* x, const float* pos, const float* properties, int flags,
                         const float* d_atom, const float* d_pos, const float* d_prop,
                         const void* fwd_workspace, void* workspace, size_t workspace_bytes, void* stream,
                         const ligdiff_dropout* dropout, const ligdiff_options* options);

/* Measurement hook (no counterpart in the reference): runs `reps` backward passes with the arguments
 * of ligdiff_backward (gradients keep accumulating into `grads`: pass throw-away buffers) and
 * returns in *mean_ms_host (host) the mean duration, from HIP events on `stream` net of an empty
 * event pair, of the wide_dw_index-th launch (in launch order within one pass) of the 256-column
 * weight-gradient kernel, and in dims3_host (host int64[3]) its shape {nout, kin, rows}:
 * dW[nout, kin] += dY[rows, nout]^T X[rows, kin], 2 nout kin rows flop.  These passes keep every launch
 * on `stream` (a training step runs the weight-gradient launches on a side stream, overlapped with the
 * input-gradient chain: there a launch's duration depends on what it shares the chip with).  Creates two
 * events and synchronises the stream; not for the training path. */
int ligdiff_backward_timed(const float* const* weights, float* const* grads, const ligdiff_graph* graph,
                           const float* x, const float* pos, const float* properties, int flags,
                           const float* d_atom, const float* d_pos, const float* d_prop,
                           const void* fwd_workspace, void* workspace, size_t workspace_bytes, void* stream,
                           int32_t wide_dw_index, int32_t reps, float* mean_ms_host, int64_t* dims3_host);

/* ---------------------------------------------------------------------------------
 * EDM noise / loss step.  Replaces ResearchAccurateTrainer._compute_research_exact_loss
 * (train.py:267-452) around the model call.
 * ------------------------------------------------------------------------------- */

/* train.py:286-312 with the uniform draws u [B] supplied by the caller:
 * sigma = (smax^(1/7) + u (smin^(1/7) - smax^(1/7)))^7; coef [B,4] = (c_skip, c_out, c_in,
 * loss weight) (model1.py:27-45); c_noise [B] = ln(sigma)/4 (the model's time input);
 * noisy = pos + sigma[mol] noise; pre_pos = noisy * c_in[mol] (the model's pos input). */
int ligdiff_edm_prepare(const float* u, const float* pos, const float* noise, const int32_t* atom_mol,
                        int64_t n_atoms, int32_t n_mols, float sigma_min, float sigma_max,
                        float sigma_data, float* coef, float* c_noise, float* noisy, float* pre_pos,
                        void* stream);

typedef struct ligdiff_loss_args {
  const float* pos;         /* [N,3] clean coordinates                                   */
  const float* noise;       /* [N,3]                                                      */
  const float* noisy;       /* [N,3] from ligdiff_edm_prepare                             */
  const float* coef;        /* [B,4] from ligdiff_edm_prepare                             */
  const float* atom_pred;   /* [N,119]                                                    */
  const float* pos_pred;    /* [N,3]                                                      */
  const float* prop_pred;   /* [B,15] (needed when properties != NULL)                    */
  const float* properties;  /* [B,15] after null-conditioning masking, or NULL            */
  const int32_t* mol_ptr;   /* [B+1]                                                      */
  int32_t n_mols;           /* local molecules                                            */
  int32_t rot_mols;         /* how many LEADING local molecules are among the global      */
                            /* batch's first two (train.py:642); 0..2                     */
  int64_t n_atoms_global;   /* atoms / molecules of the whole (all-rank) batch: the       */
  int64_t n_mols_global;    /* denominators of every mean                                 */
  void* workspace;          /* ligdiff_loss_workspace_bytes(n_mols)                       */
} ligdiff_loss_args;

size_t ligdiff_loss_workspace_bytes(int32_t n_mols);

/* sums8 [8] (device): this rank's un-normalised sums of edm_coordinate, edm_preconditioning,
 * pilot_property, dit_attention, dit_adaptive_norm, e3_coordinate, e3_rotation,
 * e3_translation (train.py:454-473, :475-498, :603-623, :625-663).  A data-parallel
 * caller all-reduces (sum) sums8 before ligdiff_loss_finish. */
int ligdiff_loss_partial(const ligdiff_loss_args* args, float* sums8, void* stream);

/* losses32 (device, 32 floats): [0..7] the eight terms as the reference reports them,
 * [8] total_loss (train.py:430-437: weighted sum, NaN/Inf terms skipped), [16..23] the
 * weight each term entered the total with.  Also writes upstream * d(total)/d(prediction)
 * into d_atom [N,119], d_pos [N,3], d_prop [B,15] (d_prop may be NULL). */
int ligdiff_loss_finish(const ligdiff_loss_args* args, const float* sums8, float upstream,
                        float* losses32, float* d_atom, float* d_pos, float* d_prop, void* stream);

/* ---------------------------------------------------------------------------------
 * Optimiser step.  Replaces torch.nn.utils.clip_grad_norm_(params, max_norm) + AdamW.step() of train.py:196-197
 * for parameters whose gradients are slices of ONE flat fp32 buffer (what the backward pass of this library's
 * Python binding produces): one reduction for the global norm and one multi-tensor update launch.
 *   table (device): n tensors x { float* param; float* exp_avg; float* exp_avg_sq; int64 grad_offset (elements into
 *   flat_grad); int64 numel };  chunk_tensor / chunk_start (device int32 [n_chunks]): workgroup c updates elements
 *   [2048 chunk_start[c], +2048) of tensor chunk_tensor[c].
 * compute_norm != 0: norm2[0] = ||flat_grad||_2 over all flat_numel elements, norm2[1] = min(1, max_norm / (norm + 1e-6))
 * (1 if max_norm <= 0); compute_norm == 0 reuses norm2 as it stands (a second group of tensors at another step count).
 * Every gradient element of the listed tensors is multiplied by norm2[1] in place (as clip_grad_norm_ leaves it), then
 *   p *= 1 - lr wd;  m += (g - m)(1 - beta1);  v = v beta2 + g^2 (1 - beta2);
 *   p -= lr / bias_correction1 * m / (sqrt(v) / bias_correction2_sqrt + eps)          (torch.optim.AdamW, no amsgrad).
 * ------------------------------------------------------------------------------- */
size_t ligdiff_adamw_workspace_bytes(void);
int ligdiff_adamw_step(const void* table_dev, const int32_t* chunk_tensor_dev, const int32_t* chunk_start_dev,
                       int32_t n_chunks, float* flat_grad, int64_t flat_numel, float lr, float beta1, float beta2,
                       float eps, float weight_decay, float bias_correction1, float bias_correction2_sqrt,
                       float max_norm, int32_t compute_norm, float* norm2_dev, void* workspace,
                       size_t workspace_bytes, void* stream);

/* Process-wide switch for how the forward pass (and the sampling loops built on it), and the
 * input-gradient products of ligdiff_backward, multiply their fp32 matrices; returns the previous setting.  0 (default, or LIGDIFF_GEMM_SPLIT unset in the
 * environment): v_mfma_f32_32x32x2_f32.  1: each fp32 operand is the exact sum of three bf16 numbers
 * and six of the nine partial products run on the bf16 matrix pipe with fp32 accumulation; the three
 * dropped products are below 2^-23 of the product, i.e. one fp32 rounding, so results agree with mode
 * 0 to fp32 rounding error.  The *_workspace_bytes functions depend on the setting (mode 1 keeps
 * split copies of the weights, about 40 MB, in the workspace): size the workspace after switching,
 * and do not switch while a call is running.  No counterpart in the reference (torch.nn.Linear in
 * fp32 throughout, model1.py:60-114, :255-302). */
int ligdiff_set_split_gemm(int enabled);

/* Tells the library that the caller is about to run k (1..16) independent calls concurrently on k streams of one
 * GPU (e.g. the halves of a sampling batch from two host threads: molecules are independent, generation.py:38-160
 * samples them one at a time).  Only the choice between the fused chain kernels and the separate launches looks at it
 * (a launch is judged on k times its own workgroups); results do not depend on it beyond that choice.  Process-wide
 * like the GEMM mode; returns the previous value, -1 for k out of range.  No counterpart in the reference. */
int ligdiff_set_concurrent_calls(int k);

/* Choice between the fused chain kernels and the separate launches on the inference path: 0 never, 1 by how well a
 * launch fills the chip, 2 always, -1 (initial state) as LIGDIFF_CHAIN in the environment says.  Process-wide like the
 * GEMM mode; returns the previous setting, -2 for a mode out of range.  Both forms compute the same layers
 * (model1.py:60-114, :255-302, :340-410) and agree to rounding.  No counterpart in the reference: measurement and A/B
 * runs use it (a label of ligdiff_forward_timed exists only in the form that launches it). */
int ligdiff_set_chain_mode(int mode);

/* Byte offsets, inside the workspace of a completed ligdiff_forward call with the same sizes and
 * flags, of the final hidden state h [n_atoms,256] (model1.py:738) and of pos + pos_updates
 * [n_atoms,3] (model1.py:752): the two inputs of the consistency module. */
int ligdiff_forward_state_offsets(int64_t n_atoms, int64_t n_edges, int32_t n_mols, int flags,
                                  size_t* hidden_offset, size_t* pos_offset);
/* The same for a forward made under per-call options (the offsets depend on layer_pairs and gemm_mode). */
int ligdiff_forward_state_offsets_opt(int64_t n_atoms, int64_t n_edges, int32_t n_mols, int flags,
                                      size_t* hidden_offset, size_t* pos_offset, const ligdiff_options* options);

/* MolDiff consistency module, train-mode fourth output of the model (model1.py:750-753, :453-561),
 * batched: the reference's Python double loop over atom pairs becomes row GEMMs over all ordered
 * pairs.  pair_ptr [n_mols+1] (device, int64): exclusive prefix sum of n_m (n_m - 1); the pairs of
 * molecule m are ordered (i, j != i) row-major as the loop at model1.py:513-540 visits them.
 * Outputs: atom_logits [N,119]; bond_logits [n_pairs,5] (may be NULL when n_pairs == 0);
 * valency [N,N] with valency[i,j] = min(valency_pred[j], max_valences[argmax_i]) - the [N] x [N,1]
 * broadcast of model1.py:468; adjacency [n_mols, n_max, n_max, 5]: softmax bond probabilities,
 * [1,0,0,0,0] on the diagonal of real atoms, zeros in the padding.  No dropout (see ligdiff_consistency_train). */
size_t ligdiff_consistency_workspace_bytes(int64_t n_atoms, int64_t n_pairs);
int ligdiff_consistency(const float* const* weights, const ligdiff_graph* graph, const float* h, const float* pos,
                        const int64_t* pair_ptr, int64_t n_pairs, int32_t n_max, float* atom_logits,
                        float* bond_logits, float* valency, float* adjacency, void* workspace,
                        size_t workspace_bytes, void* stream);

/* ligdiff_consistency with the module's train-mode dropout (dropout->p_consistency; NULL = none). */
int ligdiff_consistency_train(const float* const* weights, const ligdiff_graph* graph, const float* h, const float* pos,
                              const int64_t* pair_ptr, int64_t n_pairs, int32_t n_max, float* atom_logits,
                              float* bond_logits, float* valency, float* adjacency, void* workspace,
                              size_t workspace_bytes, void* stream, const ligdiff_dropout* dropout);

/* Measurement hook (no counterpart in the reference): runs `reps` forwards and returns in
 * *mean_ms_host (host pointer) the mean duration, from HIP events recorded on `stream`, of
 * the one launch labelled `label` (e.g. "dit1.ffn_up", "e3_1.aggregate", "dit0.qkv"),
 * net of the cost of an empty event pair measured on the same stream.
 * Creates two events and synchronises the stream; not for the serving path. */
int ligdiff_forward_timed(const float* const* weights, const ligdiff_graph* graph, const float* x,
                          const float* pos, const float* t, const int32_t* t_index_dev,
                          const float* properties, int flags, float* atom_pred, float* pos_pred,
                          float* prop_pred, void* workspace, size_t workspace_bytes, void* stream,
                          const char* label, int32_t reps, float* mean_ms_host);
/* The same under per-call options. */
int ligdiff_forward_timed_opt(const float* const* weights, const ligdiff_graph* graph, const float* x,
                              const float* pos, const float* t, const int32_t* t_index_dev,
                              const float* properties, int flags, float* atom_pred, float* pos_pred,
                              float* prop_pred, void* workspace, size_t workspace_bytes, void* stream,
                              const char* label, int32_t reps, float* mean_ms_host, const ligdiff_options* options);

/* ---------------------------------------------------------------------------------
 * Fused reverse-diffusion updates.  Replace the per-step tensor arithmetic of
 * generation.py:457-484 (DDPM) and generation.py:359-381 (DDIM): schedule lookup,
 * optional classifier-free combination (generation.py:442-443), Gaussian noise and the
 * posterior update of BOTH state tensors in one kernel.
 * ------------------------------------------------------------------------------- */
typedef struct ligdiff_step_args {
  float* x;                  /* [N,119] state, updated in place                         */
  float* pos;                /* [N,3]   state, updated in place                         */
  const float* eps_x;        /* [N,119] conditional (or only) prediction                */
  const float* eps_pos;      /* [N,3]                                                    */
  const float* eps_x_uncond; /* NULL, or unconditional prediction for guidance           */
  const float* eps_pos_uncond;
  float guidance_scale;
  float temperature;
  const float* betas;        /* [T] model buffers (model1.py:658-663)                    */
  const float* alphas;
  const float* alphas_cumprod;
  int32_t* t_index_dev;      /* device int32: current step t; decremented by the kernel */
  const float* noise_x;      /* NULL -> in-kernel Philox; else [N,119] N(0,1) draws     */
  const float* noise_pos;    /* NULL -> in-kernel Philox; else [N,3]                    */
  uint64_t seed;             /* Philox key                                               */
  int64_t atom_offset;       /* global index of atom 0 (keeps draws independent of how   */
                             /* molecules are sharded over GPUs)                         */
  int64_t n_atoms;
} ligdiff_step_args;

int ligdiff_ddpm_step(const ligdiff_step_args* args, void* stream);

/* DDIM: t_cur / t_next indices into alphas_cumprod; t_next < 0 means the final step. */
int ligdiff_ddim_step(const ligdiff_step_args* args, int32_t t_cur, int32_t t_next, void* stream);

/* Coordinate-only reverse update of the two alternate per-molecule samplers (atom features are
 * never updated by them):   pos <- (pos - b * eps) / d + c * z
 *   model1.py:956-966 (`sample`):   b = sqrt(1 - abar_t), d = sqrt(abar_t), c = sigma_t while the
 *                                   next timestep is > 0, else 0; nothing on the last step;
 *   train.py:855-863 (Euler, EDM):  b = sigma_t - sigma_next, d = 1,
 *                                   c = 0.1 * sigma_next * temperature; nothing on the last step.
 * eps = eps_pos, or eps_pos_uncond + guidance_scale * (eps_pos - eps_pos_uncond) when
 * eps_pos_uncond is given (model1.py:948, train.py:850).  z = noise_pos [n_atoms,3] if given,
 * else N(0,1) from the Philox stream 1 at (seed, step, 3 * atom_offset + element); no draw when
 * c == 0.  The scalars are host values: both samplers step through a host-side schedule. */
int ligdiff_pos_step(float* pos, const float* eps_pos, const float* eps_pos_uncond, float guidance_scale,
                     int64_t n_atoms, float b, float d, float c, const float* noise_pos, uint64_t seed,
                     uint32_t step, int64_t atom_offset, void* stream);

/*
 * Whole 1000-step DDPM loop of generation.py:422-484 for a packed batch of molecules.  The time
 * embedding and adaLN parameters of all n_steps timesteps are computed once up front (they depend
 * on the step index only); reverse steps (forward [+ unconditional forward] + fused update) are
 * captured four at a time into a hipGraph on `stream` and replayed, the step index living in
 * device memory.  x / pos hold the initial noise on entry and the sample on return.
 * properties NULL -> unguided.  use_graph != 0 needs a non-default stream (the legacy default
 * stream cannot be captured) and returns after the stream has drained; use_graph == 0 only
 * enqueues.  The workspace size depends on n_steps (the per-timestep tables).
 */
int ligdiff_ddpm_sample(const float* const* weights, const ligdiff_graph* graph, float* x, float* pos,
                        const float* properties, float guidance_scale, float temperature,
                        const float* betas, const float* alphas, const float* alphas_cumprod,
                        int32_t n_steps, uint64_t seed, int64_t atom_offset, int use_graph,
                        void* workspace, size_t workspace_bytes, void* stream);
size_t ligdiff_ddpm_sample_workspace_bytes(int64_t n_atoms, int64_t n_edges, int32_t n_mols, int guided,
                                           int32_t n_steps);

/* ligdiff_ddpm_sample / its workspace query under per-call options (NULL = the defaults). */
int ligdiff_ddpm_sample_opt(const float* const* weights, const ligdiff_graph* graph, float* x, float* pos,
                            const float* properties, float guidance_scale, float temperature,
                            const float* betas, const float* alphas, const float* alphas_cumprod,
                            int32_t n_steps, uint64_t seed, int64_t atom_offset, int use_graph,
                            void* workspace, size_t workspace_bytes, void* stream, const ligdiff_options* options);
size_t ligdiff_ddpm_sample_workspace_bytes_opt(int64_t n_atoms, int64_t n_edges, int32_t n_mols, int guided,
                                               int32_t n_steps, const ligdiff_options* options);

/* Whole DDIM loop of generation.py:279-381 (the generator's default mode), same structure as
 * ligdiff_ddpm_sample: timesteps [n_steps] (device int32, generation.py:305: linspace(T-1, 0,
 * n_steps) truncated) are walked front to back; the update of position i uses
 * alphas_cumprod[timesteps[i]] and alphas_cumprod[timesteps[i+1]] (the last step returns the x0
 * estimate).  reject [n_mols] (device int32, zeroed by the caller) receives 1 for every molecule
 * whose guidance-combined prediction failed the per-step gate of generation.py:351 / :518-541
 * (the loop still runs it to the end: molecules are independent).  No noise is drawn.  Workspace:
 * ligdiff_ddpm_sample_workspace_bytes with the same arguments. */
int ligdiff_ddim_sample(const float* const* weights, const ligdiff_graph* graph, float* x, float* pos,
                        const float* properties, float guidance_scale, const float* alphas_cumprod,
                        const int32_t* timesteps, int32_t n_steps, int32_t* reject, int use_graph,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ligdiff_ddim_sample under per-call options (workspace: ligdiff_ddpm_sample_workspace_bytes_opt). */
int ligdiff_ddim_sample_opt(const float* const* weights, const ligdiff_graph* graph, float* x, float* pos,
                            const float* properties, float guidance_scale, const float* alphas_cumprod,
                            const int32_t* timesteps, int32_t n_steps, int32_t* reject, int use_graph,
                            void* workspace, size_t workspace_bytes, void* stream, const ligdiff_options* options);

/* Fills out[n] with the N(0,1) draws the in-kernel Philox stream yields for
 * (seed, step, global element index offset..offset+n): lets a host reproduce a run. */
int ligdiff_philox_normal(float* out, int64_t n, uint64_t seed, uint32_t step, uint32_t stream_id,
                          int64_t offset, float scale, void* stream);

/* ---------------------------------------------------------------------------------
 * Generator-side helpers.
 * ------------------------------------------------------------------------------- */

/* Directed edges the bond template of generation.py:562-668 gives an n-atom molecule:
 * 0 (n<=1), 2(n-1) (n<=5), 2n (n<=15), 2n+4 (n>=16). */
int64_t ligdiff_template_edge_count(int32_t n_atoms);

/* Replaces create_research_connectivity_fixed (generation.py:562-668) for a whole batch:
 * writes edge_index [2,E] int64 (atom indices already offset by each molecule's first atom,
 * edges in the reference's order) and, if kind != NULL, kind[E] = 1 for ring/fusion bonds
 * (feature [0,0,0,1,1]) and 0 for chain/substituent bonds ([1,0,0,0,0]).
 * mol_ptr [B+1] int32 atoms, edge_ptr [B+1] int64 exclusive prefix of the per-molecule
 * edge counts, n_edges = edge_ptr[B].  Integer, bit-exact. */
int ligdiff_template_edges(const int32_t* mol_ptr, const int64_t* edge_ptr, int32_t n_mols,
                           int32_t max_mol_atoms, int64_t n_edges, int64_t* edge_index, uint8_t* kind,
                           void* stream);

/* Fully connected molecules (the large-ligand sampling configuration; the reference's sampler only
 * ever builds the template above, generation.py:409-416, but its model takes any edge_index,
 * model1.py:723): writes edge_index [2,E] int64 with every ordered pair (i, j != i) of each molecule
 * as a directed edge i -> j, atom indices offset by the molecule's first atom, pairs in row-major
 * (i, j) order.  mol_ptr [B+1] int32; edge_ptr [B+1] int64 exclusive prefix of n (n - 1);
 * n_edges = edge_ptr[B].  Integer, bit-exact. */
int ligdiff_full_edges(const int32_t* mol_ptr, const int64_t* edge_ptr, int32_t n_mols, int64_t n_edges,
                       int64_t* edge_index, void* stream);

/* generation.py:518-541 per molecule: reject[m] is SET to 1 (never cleared) when any
 * prediction of molecule m is NaN/Inf or above 100 in magnitude. */
int ligdiff_prediction_gate(const float* eps_x, const float* eps_pos, const int32_t* atom_mol,
                            int64_t n_atoms, int32_t* reject, void* stream);

/* generation.py:162-233 per molecule: accept[m] = 1 iff finite, softmax max-probability
 * >= 0.1 for every atom, per-axis coordinate variance in [1e-6, 100] and every pairwise
 * distance in [0.5, 20]. */
int ligdiff_validity_gate(const float* x, const float* pos, const int32_t* mol_ptr, int32_t n_mols,
                          int32_t* accept, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LIGDIFF_H */
