"""ctypes binding of libligdiff.so (declared in include/ligdiff.h).

There is no fallback: if the shared library is missing or a symbol is absent the import
of the hot path fails loudly.  Build it with ``python __graft_entry__.py`` (or
``csrc/build.sh``); the .so sits next to this file.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libligdiff.so")

c_f32p = C.POINTER(C.c_float)
c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)


class Graph(C.Structure):
    """struct ligdiff_graph"""
    _fields_ = [("n_atoms", C.c_int64), ("n_edges", C.c_int64), ("n_mols", C.c_int32),
                ("max_mol_atoms", C.c_int32), ("mol_ptr", C.c_void_p), ("atom_mol", C.c_void_p),
                ("dst_ptr", C.c_void_p), ("src", C.c_void_p), ("dst", C.c_void_p), ("perm", C.c_void_p),
                ("src_ptr", C.c_void_p), ("src_slot", C.c_void_p)]


class StepArgs(C.Structure):
    """struct ligdiff_step_args"""
    _fields_ = [("x", C.c_void_p), ("pos", C.c_void_p), ("eps_x", C.c_void_p), ("eps_pos", C.c_void_p),
                ("eps_x_uncond", C.c_void_p), ("eps_pos_uncond", C.c_void_p),
                ("guidance_scale", C.c_float), ("temperature", C.c_float),
                ("betas", C.c_void_p), ("alphas", C.c_void_p), ("alphas_cumprod", C.c_void_p),
                ("t_index_dev", C.c_void_p), ("noise_x", C.c_void_p), ("noise_pos", C.c_void_p),
                ("seed", C.c_uint64), ("atom_offset", C.c_int64), ("n_atoms", C.c_int64)]


ABI_VERSION = 2       # LIGDIFF_VERSION of include/ligdiff.h: struct layouts and table slots this binding mirrors
MAX_LAYER_PAIRS = 8   # LIGDIFF_MAX_LAYER_PAIRS: E(3) / attention layers the weights table has slots for


class Dropout(C.Structure):
    """struct ligdiff_dropout"""
    _fields_ = [("seed", C.c_uint64), ("atom_offset", C.c_int64), ("p_attention", C.c_float * MAX_LAYER_PAIRS),
                ("p_ffn_hidden", C.c_float * MAX_LAYER_PAIRS), ("p_ffn_out", C.c_float * MAX_LAYER_PAIRS),
                ("p_consistency", C.c_float * 2)]


class Options(C.Structure):
    """struct ligdiff_options: per-call choices (-1, or 0 for concurrent_calls, = not set)."""
    _fields_ = [("gemm_mode", C.c_int32), ("chain_mode", C.c_int32), ("concurrent_calls", C.c_int32),
                ("edge_chain", C.c_int32), ("node_chain", C.c_int32), ("mol_wave", C.c_int32),
                ("edge_reduce", C.c_int32), ("cfg_two_pass", C.c_int32), ("layer_pairs", C.c_int32)]

    def __init__(self, **fields):
        super().__init__()
        for name, _ in self._fields_:
            setattr(self, name, 0 if name == "concurrent_calls" else -1)
        for k, v in fields.items():
            if k not in dict(self._fields_):
                raise TypeError(f"ligdiff_options has no field {k}")
            setattr(self, k, int(v))

    def copy(self, **fields):
        out = Options(**{name: getattr(self, name) for name, _ in self._fields_})
        for k, v in fields.items():
            setattr(out, k, int(v))
        return out

    def as_dict(self):
        return {name: int(getattr(self, name)) for name, _ in self._fields_}


class LossArgs(C.Structure):
    """struct ligdiff_loss_args"""
    _fields_ = [("pos", C.c_void_p), ("noise", C.c_void_p), ("noisy", C.c_void_p), ("coef", C.c_void_p),
                ("atom_pred", C.c_void_p), ("pos_pred", C.c_void_p), ("prop_pred", C.c_void_p),
                ("properties", C.c_void_p), ("mol_ptr", C.c_void_p), ("n_mols", C.c_int32),
                ("rot_mols", C.c_int32), ("n_atoms_global", C.c_int64), ("n_mols_global", C.c_int64),
                ("workspace", C.c_void_p)]


# name -> (restype, argtypes); every symbol include/ligdiff.h declares
SIGNATURES = {
    "ligdiff_version": (C.c_int, []),
    "ligdiff_status_string": (C.c_char_p, [C.c_int]),
    "ligdiff_weight_count": (C.c_int, []),
    "ligdiff_weight_name": (C.c_char_p, [C.c_int]),
    "ligdiff_weight_numel": (C.c_int64, [C.c_int]),
    "ligdiff_pack_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "ligdiff_pack_csr": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32] + [C.c_void_p] * 9
                         + [C.c_void_p, C.c_size_t, C.c_void_p]),
    "ligdiff_pack_csr_static": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32] + [C.c_void_p] * 9
                                + [C.c_void_p]),
    "ligdiff_forward_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int32, C.c_int]),
    "ligdiff_forward": (C.c_int, [C.c_void_p, C.POINTER(Graph), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_size_t, C.c_void_p]),
    "ligdiff_forward_train": (C.c_int, [C.c_void_p, C.POINTER(Graph), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_size_t, C.c_void_p, C.POINTER(Dropout)]),
    "ligdiff_backward_train": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Graph), C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                         C.c_void_p, C.POINTER(Dropout)]),
    "ligdiff_options_init": (None, [C.POINTER(Options)]),
    "ligdiff_default_options": (None, [C.POINTER(Options)]),
    "ligdiff_plan": (C.c_int, [C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.POINTER(Options), C.POINTER(Options)]),
    "ligdiff_forward_workspace_bytes_opt": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int32, C.c_int, C.POINTER(Options)]),
    "ligdiff_forward_opt": (C.c_int, [C.c_void_p, C.POINTER(Graph), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_size_t, C.c_void_p, C.POINTER(Dropout), C.POINTER(Options)]),
    "ligdiff_forward_timed_opt": (C.c_int, [C.c_void_p, C.POINTER(Graph), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_size_t, C.c_void_p, C.c_char_p, C.c_int32, C.POINTER(C.c_float), C.POINTER(Options)]),
    "ligdiff_backward_workspace_bytes_opt": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int32, C.POINTER(Options)]),
    "ligdiff_backward_opt": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Graph), C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                       C.c_void_p, C.POINTER(Dropout), C.POINTER(Options)]),
    "ligdiff_ddpm_sample_opt": (C.c_int, [C.c_void_p, C.POINTER(Graph), C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                          C.c_uint64, C.c_int64, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p,
                                          C.POINTER(Options)]),
    "ligdiff_ddpm_sample_workspace_bytes_opt": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int32, C.c_int, C.c_int32,
                                                             C.POINTER(Options)]),
    "ligdiff_ddim_sample_opt": (C.c_int, [C.c_void_p, C.POINTER(Graph), C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                          C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t,
                                          C.c_void_p, C.POINTER(Options)]),
    "ligdiff_dropout_mask": (C.c_int, [C.POINTER(Dropout), C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "ligdiff_backward_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int32]),
    "ligdiff_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Graph), C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                   C.c_void_p]),
    "ligdiff_backward_timed": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Graph), C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_size_t, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_float),
                                         C.POINTER(C.c_int64)]),
    "ligdiff_adamw_workspace_bytes": (C.c_size_t, []),
    "ligdiff_adamw_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64] + [C.c_float] * 8
                           + [C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ligdiff_edm_prepare": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                      C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    "ligdiff_loss_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "ligdiff_loss_partial": (C.c_int, [C.POINTER(LossArgs), C.c_void_p, C.c_void_p]),
    "ligdiff_loss_finish": (C.c_int, [C.POINTER(LossArgs), C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "ligdiff_forward_timed": (C.c_int, [C.c_void_p, C.POINTER(Graph), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_size_t, C.c_void_p, C.c_char_p, C.c_int32, C.POINTER(C.c_float)]),
    "ligdiff_ddpm_step": (C.c_int, [C.POINTER(StepArgs), C.c_void_p]),
    "ligdiff_ddim_step": (C.c_int, [C.POINTER(StepArgs), C.c_int32, C.c_int32, C.c_void_p]),
    "ligdiff_set_split_gemm": (C.c_int, [C.c_int]),
    "ligdiff_set_concurrent_calls": (C.c_int, [C.c_int]),
    "ligdiff_set_chain_mode": (C.c_int, [C.c_int]),
    "ligdiff_forward_state_offsets": (C.c_int, [C.c_int64, C.c_int64, C.c_int32, C.c_int, C.POINTER(C.c_size_t),
                                                C.POINTER(C.c_size_t)]),
    "ligdiff_forward_state_offsets_opt": (C.c_int, [C.c_int64, C.c_int64, C.c_int32, C.c_int, C.POINTER(C.c_size_t),
                                                C.POINTER(C.c_size_t), C.POINTER(Options)]),
    "ligdiff_consistency_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "ligdiff_consistency": (C.c_int, [C.c_void_p, C.POINTER(Graph), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                      C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_size_t, C.c_void_p]),
    "ligdiff_consistency_train": (C.c_int, [C.c_void_p, C.POINTER(Graph), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                            C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_size_t, C.c_void_p, C.POINTER(Dropout)]),
    "ligdiff_pos_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_float, C.c_float,
                                   C.c_float, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int64, C.c_void_p]),
    "ligdiff_ddim_sample": (C.c_int, [C.c_void_p, C.POINTER(Graph), C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                      C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t,
                                      C.c_void_p]),
    "ligdiff_ddpm_sample_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int32, C.c_int, C.c_int32]),
    "ligdiff_ddpm_sample": (C.c_int, [C.c_void_p, C.POINTER(Graph), C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                      C.c_uint64, C.c_int64, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ligdiff_philox_normal": (C.c_int, [C.c_void_p, C.c_int64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int64,
                                        C.c_float, C.c_void_p]),
    "ligdiff_template_edge_count": (C.c_int64, [C.c_int32]),
    "ligdiff_template_edges": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    "ligdiff_full_edges": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "ligdiff_prediction_gate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "ligdiff_validity_gate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
}

FWD_PROPERTIES, FWD_NULL_COND, FWD_SKIP_PROP_HEAD, FWD_SAVE_FOR_BACKWARD, FWD_T_BROADCAST = 1, 2, 4, 8, 16

_lib = None


class LigdiffError(RuntimeError):
    pass


def load():
    """Loads libligdiff.so and binds every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LigdiffError(f"{LIB_PATH} not found: build the HIP library first (python __graft_entry__.py); "
                           "there is no CPU fallback for the hot path")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.ligdiff_version() != ABI_VERSION:
        raise LigdiffError(f"libligdiff.so has ABI version {lib.ligdiff_version()}, this binding expects {ABI_VERSION}: "
                           "rebuild it (python -c 'import __graft_entry__ as g; g.build()')")
    _lib = lib
    return lib


def check(status, what=""):
    if status != 0:
        msg = load().ligdiff_status_string(status).decode()
        raise LigdiffError(f"{what}: {msg} (status {status})")
