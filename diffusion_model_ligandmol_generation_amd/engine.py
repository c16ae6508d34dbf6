"""Host-side plumbing between PyTorch tensors and the C-ABI library.

PyTorch owns every buffer (weights, states, workspaces); this module only turns them into
raw device pointers, packs ragged batches into the CSR layout and enqueues the library's
entry points on the current HIP stream.
"""
import ctypes as C
import math
import re
import threading

import torch

from . import _lib
from ._lib import Graph, Options, check


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _opt(options):
    """ctypes argument for an optional ligdiff_options struct."""
    return C.byref(options) if options is not None else None


def _require_cuda(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on a HIP device: the hot path has no CPU implementation")


class PackedGraph:
    """Device-resident CSR form of a ragged batch (struct ligdiff_graph).

    Built from the reference's own batch layout: ``edge_index`` [2,E] int64 with
    edge_index[1] the aggregation target (model1.py:88-100) and ``batch`` [N] int64 sorted
    ascending (PyG collate, main.py:205-212).
    """

    def __init__(self, edge_index, batch, num_mols=None):
        lib = _lib.load()
        _require_cuda(batch, "batch")
        dev = batch.device
        batch = batch.contiguous().to(torch.int64)
        N = batch.numel()
        if N == 0:
            raise ValueError("empty batch")
        if edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise ValueError("edge_index must be [2, E] (a molecule without bonds needs an empty [2, 0] tensor)")
        edge_index = edge_index.contiguous().to(device=dev, dtype=torch.int64)
        E = edge_index.shape[1]
        B = int(batch.max().item()) + 1 if num_mols is None else int(num_mols)   # model1.py:702
        i32 = dict(dtype=torch.int32, device=dev)
        self.mol_ptr = torch.empty(B + 1, **i32)
        self.atom_mol = torch.empty(N, **i32)
        self.dst_ptr = torch.empty(N + 1, **i32)
        self.src_ptr = torch.empty(N + 1, **i32)
        self.src = torch.empty(max(E, 1), **i32)
        self.dst = torch.empty(max(E, 1), **i32)
        self.perm = torch.empty(max(E, 1), **i32)
        self.src_slot = torch.empty(max(E, 1), **i32)
        status = torch.zeros(2, **i32)
        ws_bytes = lib.ligdiff_pack_workspace_bytes(N, E)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = lib.ligdiff_pack_csr(_ptr(edge_index), E, _ptr(batch), N, B, _ptr(self.mol_ptr), _ptr(self.atom_mol),
                                      _ptr(self.dst_ptr), _ptr(self.src), _ptr(self.dst), _ptr(self.perm),
                                      _ptr(self.src_ptr), _ptr(self.src_slot), _ptr(status), _ptr(ws), ws_bytes,
                                      _stream())
        check(rc, "ligdiff_pack_csr")
        st = status.tolist()
        check(st[0], "ligdiff_pack_csr (device validation)")
        self.n_atoms, self.n_edges, self.n_mols, self.max_mol_atoms = N, E, B, int(st[1])
        self.device = dev
        self.batch = batch
        self.c = Graph(N, E, B, self.max_mol_atoms, self.mol_ptr.data_ptr(), self.atom_mol.data_ptr(),
                       self.dst_ptr.data_ptr(), self.src.data_ptr(), self.dst.data_ptr(), self.perm.data_ptr(),
                       self.src_ptr.data_ptr(), self.src_slot.data_ptr())


def time_frequencies(hidden_dim=256):
    """The 128 Fourier frequencies of model1.py:688-690: exp(arange * -(ln 1e4 / 127)).
    The argument is formed in fp32 as the reference does; the exponential is evaluated in
    fp64 and rounded once, so the table is the same on every host CPU (a vectorised fp32
    exp differs between machines in the last bit, which a step index near 1000 would
    amplify into a visible phase difference)."""
    half = hidden_dim // 2
    emb = math.log(10000) / (half - 1)
    arg = torch.arange(half, dtype=torch.float32) * -emb
    return torch.exp(arg.double()).float()


class ScoreNetworkEngine:
    """Weights table + workspace cache of one model on one device."""

    def __init__(self, model):
        self.lib = _lib.load()
        params = dict(model.state_dict(keep_vars=True))
        any_p = params["time_mlp.0.weight"]
        _require_cuda(any_p, "model")
        self.device = any_p.device
        self.freqs = time_frequencies(model.hidden_dim).to(self.device)
        n = self.lib.ligdiff_weight_count()
        self.names = [self.lib.ligdiff_weight_name(i).decode() for i in range(n)]
        # E(3) layers = attention layers = num_layers // 2 (model1.py:627-634); the table has slots for
        # _lib.MAX_LAYER_PAIRS of each, those of layers the model does not have stay NULL
        self.layer_pairs = len(model.dit_layers)
        if len(model.equivariant_layers) != self.layer_pairs or not 1 <= self.layer_pairs <= _lib.MAX_LAYER_PAIRS:
            raise RuntimeError("the HIP library runs 1..8 E(3) layers and as many attention layers")
        self.present = []
        self.table = (C.c_void_p * n)()
        self._sig = []
        for i, name in enumerate(self.names):
            m = re.match(r"(equivariant_layers|dit_layers)\.(\d+)\.", name)
            if m and int(m.group(2)) >= self.layer_pairs:
                self.present.append(False)
                self.table[i] = None
                self._sig.append(None)
                continue
            self.present.append(True)
            t = self.freqs if name == "__time_freqs__" else params[name]
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
                raise RuntimeError(f"weight {name} must be a contiguous fp32 tensor on {self.device}")
            if t.numel() != self.lib.ligdiff_weight_numel(i):
                raise RuntimeError(f"weight {name} has {t.numel()} elements, expected {self.lib.ligdiff_weight_numel(i)}")
            self.table[i] = t.data_ptr()
            self._sig.append(t.data_ptr())
        self.live_names = [nm for nm, ok in zip(self.names, self.present) if ok]   # slots this model fills
        self._params = params
        self._tls = threading.local()   # workspace cache and side stream of the calling thread
        self.max_atoms = model.max_atoms

    def stale(self, model):
        sd = model.state_dict(keep_vars=True)
        for i, name in enumerate(self.names):
            if self.present[i] and name != "__time_freqs__" and sd[name].data_ptr() != self._sig[i]:
                return True
        return False

    def _o(self, options):
        """ctypes argument for a call's options: the caller's struct with this model's layer count filled in
        (None, i.e. the library defaults, for a six-layer model without options)."""
        if self.layer_pairs == 3 and (options is None or options.layer_pairs in (-1, 0, 3)):
            return _opt(options)
        o = options.copy() if options is not None else Options()
        o.layer_pairs = self.layer_pairs
        return C.byref(o)

    def set_split_gemm(self, enabled):
        """Switch the library's forward GEMMs between the fp32 matrix pipe (False, default) and the
        three-way bf16 operand split on the bf16 pipe (True); returns the previous setting.
        Process-wide (include/ligdiff.h: ligdiff_set_split_gemm)."""
        return bool(self.lib.ligdiff_set_split_gemm(1 if enabled else 0))

    def ddpm_workspace_bytes(self, graph, guided, n_steps, options=None):
        return int(self.lib.ligdiff_ddpm_sample_workspace_bytes_opt(graph.n_atoms, graph.n_edges, graph.n_mols,
                                                                    1 if guided else 0, int(n_steps), self._o(options)))

    @staticmethod
    def options(**fields):
        """A ligdiff_options struct with the given fields set (gemm_mode, chain_mode, concurrent_calls, edge_chain,
        node_chain, mol_wave, edge_reduce, cfg_two_pass); everything else stays "not set"."""
        return Options(**fields)

    def default_options(self):
        """What a library call without options currently uses (process-wide setters / environment); read-only."""
        out = Options()
        self.lib.ligdiff_default_options(C.byref(out))
        return out

    def plan(self, n_atoms, n_edges, n_mols, max_mol_atoms, options=None):
        """ligdiff_plan: the kernel forms the library would pick for ONE call over a request of these sizes,
        as an Options struct to pass to every call that processes a part of the request (halves on two
        streams, shards): the parts then produce the bits of the whole, whatever their own sizes.
        Fields that ``options`` already sets are kept."""
        out = Options()
        with torch.cuda.device(self.device):
            check(self.lib.ligdiff_plan(int(n_atoms), int(n_edges), int(n_mols), int(max_mol_atoms), self._o(options),
                                        C.byref(out)), "ligdiff_plan")
        return out

    def plan_graph(self, graph, options=None):
        return self.plan(graph.n_atoms, graph.n_edges, graph.n_mols, graph.max_mol_atoms, options)

    def set_concurrent_calls(self, k):
        """Hint for k concurrent library calls on one GPU (include/ligdiff.h: ligdiff_set_concurrent_calls);
        returns the previous value."""
        prev = int(self.lib.ligdiff_set_concurrent_calls(int(k)))
        if prev < 0:
            raise ValueError("concurrent calls must be 1..16")
        return prev

    def set_chain_mode(self, mode):
        """Fused chain kernels: 0 never, 1 by chip fill, 2 always, -1 as the environment says
        (include/ligdiff.h: ligdiff_set_chain_mode); returns the previous setting."""
        prev = int(self.lib.ligdiff_set_chain_mode(int(mode)))
        if prev < -1:
            raise ValueError("chain mode must be -1..2")
        return prev

    def workspace(self, nbytes):
        """Cached workspace tensor of the calling host thread (library calls made from different threads run
        concurrently on their own streams and must not share one)."""
        ws = getattr(self._tls, "ws", None)
        if ws is None or ws.numel() < nbytes:
            ws = self._tls.ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return ws

    def forward(self, graph, x, pos, t, properties=None, null_cond=False, want_props=True, save=False,
                workspace=None, t_index=None, keep_state=False, dropout=None, options=None):
        """ligdiff_forward on the current stream.  t: [B] float tensor (or [1], applied to
        every molecule); alternatively t_index, a device int32 scalar holding the step.
        Returns atom_pred, pos_pred, prop_pred (None if not wanted)."""
        N, B = graph.n_atoms, graph.n_mols
        if t is None:
            t = torch.zeros(1, device=self.device)
        self._check_sizes(graph)
        for name, v in (("x", x), ("pos", pos), ("t", t)):
            _require_cuda(v, name)
        x = x.contiguous().float()
        pos = pos.contiguous().float()
        t = t.reshape(-1).contiguous().float()
        if x.shape != (N, 119) or pos.shape != (N, 3):
            raise ValueError("x must be [N,119] and pos [N,3]")
        flags = 0
        if t_index is not None:
            if t_index.dtype != torch.int32 or not t_index.is_cuda:
                raise ValueError("t_index must be a device int32 tensor")
            flags |= _lib.FWD_T_BROADCAST
        elif t.numel() == 1 and B > 1:
            flags |= _lib.FWD_T_BROADCAST
        elif t.numel() < B:
            raise ValueError("t needs one value per molecule")
        if properties is not None:
            flags |= _lib.FWD_PROPERTIES
            if null_cond:
                flags |= _lib.FWD_NULL_COND
            else:
                properties = properties.contiguous().float()
                if properties.shape[0] != B or properties.shape[1] < 8:
                    raise ValueError("properties must be [B, >=8] (a [B,15] tensor in the reference)")
                if properties.shape[1] != 15:
                    p15 = properties.new_zeros(B, 15)
                    p15[:, :min(15, properties.shape[1])] = properties[:, :15]
                    properties = p15
        if not want_props:
            flags |= _lib.FWD_SKIP_PROP_HEAD
        if save:
            flags |= _lib.FWD_SAVE_FOR_BACKWARD
        nbytes = self.lib.ligdiff_forward_workspace_bytes_opt(N, graph.n_edges, B, flags, self._o(options))
        ws = workspace if workspace is not None else self.workspace(nbytes)
        atom_pred = torch.empty(N, 119, dtype=torch.float32, device=self.device)
        pos_pred = torch.empty(N, 3, dtype=torch.float32, device=self.device)
        prop_pred = torch.empty(B, 15, dtype=torch.float32, device=self.device) if want_props else None
        with torch.cuda.device(self.device):
            rc = self.lib.ligdiff_forward_opt(self.table, C.byref(graph.c), _ptr(x), _ptr(pos), _ptr(t), _ptr(t_index),
                                              _ptr(properties) if (properties is not None and not null_cond) else None,
                                              flags, _ptr(atom_pred), _ptr(pos_pred), _ptr(prop_pred), _ptr(ws),
                                              ws.numel(), _stream(),
                                              C.byref(dropout) if dropout is not None else None, self._o(options))
        check(rc, "ligdiff_forward")
        self.last_forward = (ws, flags) if keep_state else None   # for consistency() right after this call
        return atom_pred, pos_pred, prop_pred

    def _check_sizes(self, graph):
        """model1.py:728 pads every molecule to max_atoms: a larger molecule fails there (the reference
        then returns zeros, model1.py:757-764); here it raises."""
        if graph.max_mol_atoms > self.max_atoms:
            raise RuntimeError(f"a molecule has {graph.max_mol_atoms} atoms but the model was built with "
                               f"max_atoms={self.max_atoms} (model1.py:728)")

    def time_launch(self, label, graph, x, pos, t_index, reps=5, options=None):
        """Mean duration in ms (HIP events on the launch stream) of one labelled launch of
        the forward pass in situ; measurement hook used by bench.py."""
        flags = _lib.FWD_T_BROADCAST | _lib.FWD_SKIP_PROP_HEAD
        nbytes = self.lib.ligdiff_forward_workspace_bytes_opt(graph.n_atoms, graph.n_edges, graph.n_mols, flags, self._o(options))
        ws = self.workspace(nbytes)
        a = torch.empty(graph.n_atoms, 119, dtype=torch.float32, device=self.device)
        p = torch.empty(graph.n_atoms, 3, dtype=torch.float32, device=self.device)
        ms = C.c_float(0.0)
        with torch.cuda.device(self.device):
            rc = self.lib.ligdiff_forward_timed_opt(self.table, C.byref(graph.c), _ptr(x), _ptr(pos), None, _ptr(t_index),
                                                    None, flags, _ptr(a), _ptr(p), None, _ptr(ws), ws.numel(), _stream(),
                                                    label.encode(), int(reps), C.byref(ms), self._o(options))
        check(rc, f"ligdiff_forward_timed({label})")
        return float(ms.value)

    def dropout_mask(self, dropout, layer, kind, n_atoms):
        """The mask scales (0 or 1 / (1 - p)) the library applies at one dropout site for atoms
        [0, n_atoms) of a call made with ``dropout`` (ligdiff_dropout_mask): kind 0 attention
        probabilities [n_atoms, 8, 128] (query atom, head, key index in its molecule), 1 FFN hidden
        [n_atoms, 1024], 2 FFN output [n_atoms, 256], 3 / 4 the consistency module's atom-type / bond predictor
        hidden [n_atoms or ordered pairs, 128]."""
        shape = {0: (n_atoms, 8, 128), 1: (n_atoms, 1024), 2: (n_atoms, 256), 3: (n_atoms, 128), 4: (n_atoms, 128)}[kind]
        out = torch.empty(shape, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.ligdiff_dropout_mask(C.byref(dropout), int(layer), int(kind), int(n_atoms), _ptr(out), _stream()),
                  "ligdiff_dropout_mask")
        return out

    def time_backward_launch(self, graph, x, pos, t, wide_dw_index, reps=5):
        """Mean duration in ms of one wide weight-gradient launch of the backward pass in situ
        (ligdiff_backward_timed) and its shape (nout, kin, rows); measurement hook used by bench.py.
        Runs a forward with saved activations and ``reps`` backward passes into throw-away gradients."""
        lib, dev = self.lib, self.device
        N, B = graph.n_atoms, graph.n_mols
        flags = _lib.FWD_SAVE_FOR_BACKWARD
        fws = torch.empty(lib.ligdiff_forward_workspace_bytes(N, graph.n_edges, B, flags), dtype=torch.uint8, device=dev)
        atom, posp = torch.empty(N, 119, device=dev), torch.empty(N, 3, device=dev)
        prop = torch.empty(B, 15, device=dev)
        t = t.reshape(-1).contiguous().float()
        grads = [torch.zeros(lib.ligdiff_weight_numel(i), device=dev) for i in range(len(self.names))]
        gtable = (C.c_void_p * len(self.names))(*[g.data_ptr() for g in grads])
        d_atom, d_pos = torch.randn(N, 119, device=dev) * 1e-3, torch.randn(N, 3, device=dev) * 1e-3
        bws = torch.empty(lib.ligdiff_backward_workspace_bytes(N, graph.n_edges, B), dtype=torch.uint8, device=dev)
        ms, dims = C.c_float(0.0), (C.c_int64 * 3)()
        with torch.cuda.device(dev):
            check(lib.ligdiff_forward(self.table, C.byref(graph.c), _ptr(x), _ptr(pos), _ptr(t), None, None, flags,
                                      _ptr(atom), _ptr(posp), _ptr(prop), _ptr(fws), fws.numel(), _stream()),
                  "ligdiff_forward(save)")
            check(lib.ligdiff_backward_timed(self.table, gtable, C.byref(graph.c), _ptr(x), _ptr(pos), None, flags,
                                             _ptr(d_atom), _ptr(d_pos), None, _ptr(fws), _ptr(bws), bws.numel(),
                                             _stream(), int(wide_dw_index), int(reps), C.byref(ms), dims),
                  "ligdiff_backward_timed")
        return float(ms.value), tuple(int(v) for v in dims)

    # ------------------------------------------------------------------ sampler entry points
    def step_args(self, model, x, pos, eps_x, eps_pos, eps_xu=None, eps_pu=None, guidance_scale=1.0,
                  temperature=1.0, t_index=None, noise_x=None, noise_pos=None, seed=0, atom_offset=0):
        a = _lib.StepArgs()
        a.x, a.pos, a.eps_x, a.eps_pos = x.data_ptr(), pos.data_ptr(), eps_x.data_ptr(), eps_pos.data_ptr()
        a.eps_x_uncond = eps_xu.data_ptr() if eps_xu is not None else None
        a.eps_pos_uncond = eps_pu.data_ptr() if eps_pu is not None else None
        a.guidance_scale, a.temperature = float(guidance_scale), float(temperature)
        a.betas, a.alphas = model.betas.data_ptr(), model.alphas.data_ptr()
        a.alphas_cumprod = model.alphas_cumprod.data_ptr()
        a.t_index_dev = t_index.data_ptr() if t_index is not None else None
        a.noise_x = noise_x.data_ptr() if noise_x is not None else None
        a.noise_pos = noise_pos.data_ptr() if noise_pos is not None else None
        a.seed, a.atom_offset, a.n_atoms = int(seed), int(atom_offset), x.shape[0]
        return a

    def ddpm_step(self, args):
        with torch.cuda.device(self.device):
            check(self.lib.ligdiff_ddpm_step(C.byref(args), _stream()), "ligdiff_ddpm_step")

    def ddim_step(self, args, t_cur, t_next):
        with torch.cuda.device(self.device):
            check(self.lib.ligdiff_ddim_step(C.byref(args), int(t_cur), int(t_next), _stream()), "ligdiff_ddim_step")

    def consistency(self, graph, ws, flags, dropout=None, options=None):
        """MolDiff consistency outputs (model1.py:750-753) from the state a forward call left in
        its workspace ``ws`` (called with ``flags``).  Returns the reference's 4-tuple
        (atom_logits [N,119], bond_logits [P,5] or None, valency_constrained [N,N],
        adjacency [B,n_max,n_max,5]); none of them carries gradient."""
        lib, dev = self.lib, self.device
        N, B = graph.n_atoms, graph.n_mols
        h_off, p_off = C.c_size_t(0), C.c_size_t(0)
        check(lib.ligdiff_forward_state_offsets_opt(N, graph.n_edges, B, int(flags), C.byref(h_off), C.byref(p_off),
                                                    self._o(options)), "ligdiff_forward_state_offsets")
        sizes = (graph.mol_ptr[1:] - graph.mol_ptr[:-1]).to(torch.int64)
        pair_ptr = torch.zeros(B + 1, dtype=torch.int64, device=dev)
        pair_ptr[1:] = torch.cumsum(sizes * (sizes - 1), 0)
        n_pairs = int(pair_ptr[-1].item())
        n_max = graph.max_mol_atoms
        atom_logits = torch.empty(N, 119, dtype=torch.float32, device=dev)
        bond_logits = torch.empty(n_pairs, 5, dtype=torch.float32, device=dev) if n_pairs > 0 else None
        valency = torch.empty(N, N, dtype=torch.float32, device=dev)
        adjacency = torch.empty(B, n_max, n_max, 5, dtype=torch.float32, device=dev)
        cws = torch.empty(lib.ligdiff_consistency_workspace_bytes(N, n_pairs), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = lib.ligdiff_consistency_train(self.table, C.byref(graph.c), ws.data_ptr() + h_off.value,
                                               ws.data_ptr() + p_off.value, _ptr(pair_ptr), n_pairs, n_max,
                                               _ptr(atom_logits), _ptr(bond_logits), _ptr(valency), _ptr(adjacency),
                                               _ptr(cws), cws.numel(), _stream(),
                                               C.byref(dropout) if dropout is not None else None)
        check(rc, "ligdiff_consistency")
        self.last_forward = None
        return atom_logits, bond_logits, valency, adjacency

    def pos_step(self, pos, eps_pos, b, d, c, eps_pos_uncond=None, guidance_scale=1.0, noise_pos=None, seed=0,
                 step=0, atom_offset=0):
        """pos <- (pos - b * eps) / d + c * z in place (ligdiff_pos_step): the coordinate update of
        the alternate samplers, model1.py:956-966 and train.py:855-863."""
        with torch.cuda.device(self.device):
            check(self.lib.ligdiff_pos_step(_ptr(pos), _ptr(eps_pos), _ptr(eps_pos_uncond), float(guidance_scale),
                                            pos.shape[0], float(b), float(d), float(c), _ptr(noise_pos), int(seed),
                                            int(step), int(atom_offset), _stream()), "ligdiff_pos_step")

    def ddpm_sample(self, model, graph, x, pos, properties=None, guidance_scale=1.0, temperature=1.0,
                    n_steps=None, seed=0, atom_offset=0, use_graph=True, workspace=None, stream=None, options=None):
        """Whole reverse loop in the library (hipGraph replay); x / pos updated in place.
        workspace / stream: a caller that runs several of these calls concurrently (one host thread each)
        gives every call its own workspace tensor and its own non-default stream; the call then returns
        with the loop finished on that stream and leaves the current stream alone."""
        n_steps = model.timesteps if n_steps is None else n_steps
        self._check_sizes(graph)
        guided = 1 if (properties is not None and guidance_scale > 1.0) else 0
        nbytes = self.lib.ligdiff_ddpm_sample_workspace_bytes_opt(graph.n_atoms, graph.n_edges, graph.n_mols, guided,
                                                                  int(n_steps), self._o(options))
        if workspace is not None and workspace.numel() < nbytes:
            raise ValueError("workspace too small")
        ws = self.workspace(nbytes) if workspace is None else workspace
        if stream is not None:
            if properties is not None:
                properties = properties.contiguous().float()
            with torch.cuda.device(self.device), torch.cuda.stream(stream):
                rc = self.lib.ligdiff_ddpm_sample_opt(self.table, C.byref(graph.c), _ptr(x), _ptr(pos), _ptr(properties),
                                                      float(guidance_scale), float(temperature), _ptr(model.betas),
                                                      _ptr(model.alphas), _ptr(model.alphas_cumprod), int(n_steps),
                                                      int(seed), int(atom_offset), 1 if use_graph else 0, _ptr(ws),
                                                      ws.numel(), C.c_void_p(stream.cuda_stream), self._o(options))
            check(rc, "ligdiff_ddpm_sample")
            return
        if properties is not None:
            properties = properties.contiguous().float()
            if properties.shape != (graph.n_mols, 15):
                raise ValueError("properties must be [B,15]")
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream()
            # stream capture is not possible on the legacy default stream: hop to a side stream
            run_on = cur
            if use_graph and cur.cuda_stream == 0:
                if getattr(self._tls, "side", None) is None:
                    self._tls.side = torch.cuda.Stream(device=self.device)
                run_on = self._tls.side
                run_on.wait_stream(cur)
            with torch.cuda.stream(run_on):
                rc = self.lib.ligdiff_ddpm_sample_opt(self.table, C.byref(graph.c), _ptr(x), _ptr(pos), _ptr(properties),
                                                      float(guidance_scale), float(temperature), _ptr(model.betas),
                                                      _ptr(model.alphas), _ptr(model.alphas_cumprod), int(n_steps),
                                                      int(seed), int(atom_offset), 1 if use_graph else 0, _ptr(ws),
                                                      ws.numel(), _stream(), self._o(options))
            if run_on is not cur:
                cur.wait_stream(run_on)
        check(rc, "ligdiff_ddpm_sample")

    def ddim_sample(self, model, graph, x, pos, timesteps, reject, properties=None, guidance_scale=1.0,
                    use_graph=True, options=None):
        """Whole DDIM loop in the library (ligdiff_ddim_sample); x / pos updated in place, ``reject``
        [B] int32 receives the per-step prediction gate.  timesteps: list of ints."""
        n_steps = len(timesteps)
        self._check_sizes(graph)
        guided = 1 if (properties is not None and guidance_scale > 1.0) else 0
        nbytes = self.lib.ligdiff_ddpm_sample_workspace_bytes_opt(graph.n_atoms, graph.n_edges, graph.n_mols, guided,
                                                                  n_steps, self._o(options))
        ws = self.workspace(nbytes)
        ts = torch.tensor([int(t) for t in timesteps], dtype=torch.int32, device=self.device)
        if properties is not None:
            properties = properties.contiguous().float()
            if properties.shape != (graph.n_mols, 15):
                raise ValueError("properties must be [B,15]")
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream()
            run_on = cur
            if use_graph and cur.cuda_stream == 0:     # the legacy default stream cannot be captured
                if getattr(self._tls, "side", None) is None:
                    self._tls.side = torch.cuda.Stream(device=self.device)
                run_on = self._tls.side
                run_on.wait_stream(cur)
            with torch.cuda.stream(run_on):
                rc = self.lib.ligdiff_ddim_sample_opt(self.table, C.byref(graph.c), _ptr(x), _ptr(pos), _ptr(properties),
                                                      float(guidance_scale), _ptr(model.alphas_cumprod), _ptr(ts), n_steps,
                                                      _ptr(reject), 1 if use_graph else 0, _ptr(ws), ws.numel(), _stream(),
                                                      self._o(options))
            if run_on is not cur:
                cur.wait_stream(run_on)
        check(rc, "ligdiff_ddim_sample")

    def philox_normal(self, n, seed, step, stream_id, offset=0, scale=1.0):
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.ligdiff_philox_normal(_ptr(out), n, int(seed), int(step), int(stream_id), int(offset),
                                                 float(scale), _stream()), "ligdiff_philox_normal")
        return out

    def validity_gate(self, x, pos, mol_ptr, n_mols):
        accept = torch.empty(n_mols, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.ligdiff_validity_gate(_ptr(x), _ptr(pos), _ptr(mol_ptr), n_mols, _ptr(accept), _stream()),
                  "ligdiff_validity_gate")
        return accept

    def prediction_gate(self, eps_x, eps_pos, graph, reject):
        with torch.cuda.device(self.device):
            check(self.lib.ligdiff_prediction_gate(_ptr(eps_x), _ptr(eps_pos), _ptr(graph.atom_mol), graph.n_atoms,
                                                   _ptr(reject), _stream()), "ligdiff_prediction_gate")


def _size_arrays(sizes, counts, device):
    """mol_ptr [B+1] int32, edge_ptr [B+1] int64 (exclusive prefixes) on ``device``, batch [N] int64."""
    B = len(sizes)
    mol_ptr = torch.zeros(B + 1, dtype=torch.int32)
    mol_ptr[1:] = torch.tensor(sizes, dtype=torch.int64).cumsum(0).to(torch.int32)
    edge_ptr = torch.zeros(B + 1, dtype=torch.int64)
    edge_ptr[1:] = torch.tensor(counts, dtype=torch.int64).cumsum(0)
    batch = torch.repeat_interleave(torch.arange(B, dtype=torch.int64), torch.tensor(sizes)).to(device)
    return mol_ptr.to(device), edge_ptr.to(device), int(edge_ptr[-1]), batch


def _check_sizes_list(sizes):
    sizes = [int(n) for n in sizes]
    if len(sizes) == 0 or min(sizes) < 1:
        raise ValueError("molecule sizes must be >= 1")
    if min(sizes) < 2:
        raise ValueError("a one-atom molecule has no bonds: outside the generator's supported domain")
    return sizes


def template_graph(sizes, device):
    """Batch of molecules with the generator's bond template (generation.py:562-668), built
    on device.  Returns (PackedGraph, edge_index [2,E] int64, kind [E] uint8)."""
    lib = _lib.load()
    sizes = _check_sizes_list(sizes)
    B = len(sizes)
    counts = [lib.ligdiff_template_edge_count(n) for n in sizes]
    mol_ptr_d, edge_ptr_d, E, batch = _size_arrays(sizes, counts, device)
    edge_index = torch.empty(2, E, dtype=torch.int64, device=device)
    kind = torch.empty(E, dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        check(lib.ligdiff_template_edges(_ptr(mol_ptr_d), _ptr(edge_ptr_d), B, max(sizes), E, _ptr(edge_index),
                                         _ptr(kind), _stream()), "ligdiff_template_edges")
    return PackedGraph(edge_index, batch, num_mols=B), edge_index, kind


def full_graph(sizes, device):
    """Batch of fully connected molecules: every ordered pair (i, j != i) of a molecule is a directed
    edge (ligdiff_full_edges), built on device.  Returns (PackedGraph, edge_index [2,E] int64)."""
    lib = _lib.load()
    sizes = _check_sizes_list(sizes)
    B = len(sizes)
    mol_ptr_d, edge_ptr_d, E, batch = _size_arrays(sizes, [n * (n - 1) for n in sizes], device)
    edge_index = torch.empty(2, E, dtype=torch.int64, device=device)
    with torch.cuda.device(device):
        check(lib.ligdiff_full_edges(_ptr(mol_ptr_d), _ptr(edge_ptr_d), B, E, _ptr(edge_index), _stream()),
              "ligdiff_full_edges")
    return PackedGraph(edge_index, batch, num_mols=B), edge_index


def graph_from_edges(sizes, edge_index, device):
    """Batch whose bonds the caller supplies: ``edge_index`` [2,E] int64 holds GLOBAL atom indices
    (molecule m owns atoms sum(sizes[:m]) .. sum(sizes[:m+1]), as PyG collation numbers them).
    Returns the PackedGraph."""
    sizes = _check_sizes_list(sizes)
    batch = torch.repeat_interleave(torch.arange(len(sizes), dtype=torch.int64), torch.tensor(sizes)).to(device)
    return PackedGraph(edge_index.to(device), batch, num_mols=len(sizes))
