"""Drop-in score network: the reference's constructor / forward signature over the HIP library.

Mirrors ResearchAccurateDiffusionModel of model1.py:564-764.  The module tree only holds
parameters (see modules.py); ``forward`` packs the ragged batch into the CSR layout and
runs the whole network through the C-ABI library (csrc/), on the current HIP stream.
There is no CPU or eager-PyTorch fallback: on a machine without the library or without
a GPU, ``forward`` raises.
"""
import math

import torch
import torch.nn as nn

from .modules import (EDMPreconditioning, E3EquivariantMessagePassing, GraphDiTLayer,
                      MolDiffConsistencyModule, PILOTMultiObjectiveGuidance)


def cosine_beta_schedule(timesteps):
    """Host-side, one-off: model1.py:674-684."""
    s = 0.008
    x = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float32)
    ac = torch.cos(((x / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    return torch.clamp(1 - (ac[1:] / ac[:-1]), 0.0001, 0.02)


class ResearchAccurateDiffusionModel(nn.Module):
    """Same constructor arguments, attributes, buffers and state_dict keys as
    model1.py:574-665.  hidden_dim is fixed at 256, num_heads at 8 and atom_feature_dim at 119 by the
    kernels' tiling (other values raise); num_layers // 2 E(3) layers and as many attention layers as in
    model1.py:627-634, for num_layers from 2 to 17 (6 is the reference's default, main.py:242 passes 8)."""

    def __init__(self, atom_feature_dim=119, edge_feature_dim=5, hidden_dim=256, num_layers=6,
                 num_heads=8, timesteps=1000, property_dim=15, max_atoms=100,
                 sigma_min=0.002, sigma_max=80.0, sigma_data=0.5,
                 use_equivariance=True, use_consistency=True, use_multi_objective=True):
        super().__init__()
        # main.py:238-249 passes the three use_* switches (all True) to the model class it imports; the class of
        # model1.py has no such parameters: every component is always built.  Accepted here as long as they ask for that.
        if not (use_equivariance and use_consistency and use_multi_objective):
            raise ValueError("the E(3) layers, the consistency module and the PILOT guidance are always built "
                             "(model1.py:574-665 has no switch for them)")
        if hidden_dim != 256 or num_heads != 8 or atom_feature_dim != 119 or property_dim != 15:
            raise ValueError("the HIP kernels are built for hidden_dim=256, num_heads=8, "
                             "atom_feature_dim=119, property_dim=15")
        if not 1 <= num_layers // 2 <= 8:
            raise ValueError("the HIP library runs 1 to 8 E(3) layers and as many attention layers "
                             "(num_layers // 2 of each, model1.py:627-634): num_layers must be 2..17")
        if max_atoms > 128:
            raise ValueError("max_atoms above 128 is not supported by the attention kernel")
        self.atom_feature_dim = atom_feature_dim
        self.hidden_dim = hidden_dim
        self.timesteps = timesteps
        self.max_atoms = max_atoms
        self.num_heads = num_heads
        self.property_dim = property_dim

        self.edm_preconditioning = EDMPreconditioning(sigma_min, sigma_max, sigma_data)
        self.atom_embedding = nn.Sequential(nn.Linear(atom_feature_dim, hidden_dim), nn.LayerNorm(hidden_dim), nn.GELU())
        self.pos_embedding = nn.Sequential(nn.Linear(3, hidden_dim), nn.LayerNorm(hidden_dim), nn.GELU())
        self.time_mlp = nn.Sequential(nn.Linear(hidden_dim, hidden_dim * 2), nn.GELU(),
                                      nn.Linear(hidden_dim * 2, hidden_dim))
        self.property_guidance = PILOTMultiObjectiveGuidance(property_dim, hidden_dim)
        self.equivariant_layers = nn.ModuleList(
            [E3EquivariantMessagePassing(hidden_dim) for _ in range(num_layers // 2)])
        self.dit_layers = nn.ModuleList([GraphDiTLayer(hidden_dim, num_heads) for _ in range(num_layers // 2)])
        self.moldiff_consistency = MolDiffConsistencyModule(hidden_dim, max_atoms)

        def head(out):
            return nn.Sequential(nn.Linear(hidden_dim, hidden_dim), nn.GELU(), nn.LayerNorm(hidden_dim),
                                 nn.Linear(hidden_dim, out))

        self.atom_output = head(atom_feature_dim)
        self.pos_output = head(3)
        self.property_output = head(property_dim)

        betas = cosine_beta_schedule(timesteps)
        self.register_buffer("betas", betas)
        alphas = 1.0 - betas
        self.register_buffer("alphas", alphas)
        alphas_cumprod = torch.cumprod(alphas, dim=0)
        self.register_buffer("alphas_cumprod", alphas_cumprod)
        self.register_buffer("sqrt_alphas_cumprod", torch.sqrt(alphas_cumprod))
        self.register_buffer("sqrt_one_minus_alphas_cumprod", torch.sqrt(1.0 - alphas_cumprod))
        self._engine = None

    # ------------------------------------------------------------------ engine plumbing
    def engine(self):
        """The per-model handle on the HIP library (weights table + workspaces)."""
        from .engine import ScoreNetworkEngine
        if self._engine is None or self._engine.stale(self):
            self._engine = ScoreNetworkEngine(self)
        return self._engine

    def dropout_spec(self, atom_offset=0, seed=None):
        """The train-mode dropout of one forward / backward pair as the library takes it (struct
        ligdiff_dropout), or None in eval mode or when every probability is 0.  The probabilities are
        read from the modules the reference configures (model1.py:263-274): ``attention.dropout`` and
        the two ``nn.Dropout`` of ``ff_network``.  seed None: a fresh key from torch's global generator
        (so torch.manual_seed makes a run repeatable).  atom_offset: global index of the batch's first
        atom (a data-parallel shard passes the atoms that precede it; the mask then equals the
        unsharded batch's)."""
        if not self.training:
            return None
        from ._lib import Dropout
        d = Dropout()
        any_p = False
        for l, lyr in enumerate(self.dit_layers):
            d.p_attention[l] = float(lyr.attention.dropout)
            d.p_ffn_hidden[l] = float(lyr.ff_network[2].p)
            d.p_ffn_out[l] = float(lyr.ff_network[4].p)
            any_p = any_p or d.p_attention[l] > 0 or d.p_ffn_hidden[l] > 0 or d.p_ffn_out[l] > 0
        cons = self.moldiff_consistency                                  # model1.py:390, :401
        d.p_consistency[0], d.p_consistency[1] = float(cons.atom_predictor[5].p), float(cons.bond_predictor[5].p)
        any_p = any_p or d.p_consistency[0] > 0 or d.p_consistency[1] > 0
        if not any_p:
            return None
        d.seed = int(torch.randint(0, 1 << 62, (1,), dtype=torch.int64).item()) if seed is None else int(seed)
        d.atom_offset = int(atom_offset)
        return d

    def forward(self, data, t, properties=None, classifier_free_guidance=False):
        """model1.py:699-755.  ``data`` is any object with .x [N,119], .pos [N,3],
        .edge_index [2,E] int64 and .batch [N] int64 (sorted).  Returns
        (atom_pred [N,119], pos_pred [N,3], prop_pred [B,15], consistency_outputs).

        Train mode applies the attention layers' dropout (see ``dropout_spec``); ``data.dropout_seed``
        and ``data.atom_offset``, when present, fix the mask's key and the batch's global position.

        consistency_outputs: None in eval mode; in train mode the MolDiff consistency module's
        (atom_logits, bond_logits, valency_constrained, adjacency_matrix) as model1.py:750-753
        returns them, with the module's own dropout and detached (none of them reaches a loss with
        the data pipeline's batches: train.py:519, :543, :559, :583).  A true
        ``data.skip_consistency`` (the attribute model1.py:928 attaches) or
        ``model.skip_consistency`` leaves it out: its [N,N] valency output is quadratic in the
        batch's atom count.
        Errors raise instead of being turned into zero predictions (model1.py:757-764).
        """
        from .autograd import packed_graph_of, score_network_apply
        want_cons = (self.training and not getattr(self, "skip_consistency", False)
                     and not getattr(data, "skip_consistency", False))
        dropout = self.dropout_spec(getattr(data, "atom_offset", 0), getattr(data, "dropout_seed", None))
        atom_pred, pos_pred, prop_pred = score_network_apply(self, data, t, properties, classifier_free_guidance,
                                                             keep_state=want_cons, dropout=dropout)
        cons = None
        if want_cons:
            eng = self.engine()
            ws, flags = eng.last_forward
            with torch.no_grad():
                cons = eng.consistency(packed_graph_of(data), ws, flags, dropout, options=getattr(self, "library_options", None))
        return atom_pred, pos_pred, prop_pred, cons

    def compute_loss(self, data, t, properties=None, loss_weights=None, draws=None):
        """model1.py:767-888.  Returns (losses, metrics) with the reference's keys.  ``t`` is unused,
        as in the reference: the model is called with the sampled sigma as its time input and the
        un-preconditioned noisy coordinates (model1.py:786-805); the coordinate term is
        mean_atoms(mean_xyz((pos_pred - noise)^2) * w(sigma)) (model1.py:824-826).

        The noising runs in ligdiff_edm_prepare, forward / backward in the HIP library (losses['total']
        is differentiable w.r.t. the parameters); the scalar reductions are a handful of device
        tensor ops.  ``draws`` optionally injects the random numbers: 'u' [B] uniforms of the sigma
        draw, 'noise' [N,3], 'null_rand' [B] uniforms of the null-conditioning mask.
        The label-gated consistency terms (model1.py:842-886) are evaluated when the batch carries
        the labels and the model is in train mode; the consistency outputs here carry no gradient."""
        import ctypes as C
        import torch.nn.functional as F
        from ._lib import check
        from .autograd import packed_graph_of
        from .engine import _ptr, _stream
        if loss_weights is None:
            loss_weights = {"coordinate": 1.0, "atom_type": 0.5, "property": 0.3, "valency": 0.4, "adjacency": 0.6}
        eng = self.engine()
        dev = eng.device
        if not hasattr(data, "batch") or data.batch is None:
            data.batch = torch.zeros(data.x.shape[0], dtype=torch.long, device=data.x.device)
        graph = packed_graph_of(data)
        N, B = graph.n_atoms, graph.n_mols
        draws = draws or {}
        edm = self.edm_preconditioning
        u = draws["u"].to(dev).float() if "u" in draws else torch.rand(B, device=dev)          # model1.py:786
        pos = data.pos.to(dev).float().contiguous()
        noise = draws["noise"].to(dev).float().contiguous() if "noise" in draws else torch.randn_like(pos)
        sigma = edm.sigma_from_u(u)
        coef = torch.empty(B, 4, device=dev)
        c_noise = torch.empty(B, device=dev)
        noisy = torch.empty_like(pos)
        pre_pos = torch.empty_like(pos)
        with torch.cuda.device(dev):
            check(eng.lib.ligdiff_edm_prepare(_ptr(u), _ptr(pos), _ptr(noise), _ptr(graph.atom_mol), N, B,
                                              float(edm.sigma_min), float(edm.sigma_max), float(edm.sigma_data),
                                              _ptr(coef), _ptr(c_noise), _ptr(noisy), _ptr(pre_pos), _stream()),
                  "ligdiff_edm_prepare")

        class _Noisy:
            pass

        nd = _Noisy()
        nd.x, nd.pos, nd.edge_index, nd.batch, nd._ligdiff_graph = data.x, noisy, data.edge_index, data.batch, graph
        atom_pred, pos_pred, prop_pred, cons = self.forward(nd, sigma, properties)
        zero = lambda: torch.tensor(0.0, device=dev)   # noqa: E731
        losses = {k: zero() for k in ("total", "coordinate", "atom_type", "property", "valency", "adjacency")}
        metrics = {"atom_accuracy": 0.0, "valency_mae": 0.0, "adjacency_accuracy": 0.0}
        batch = graph.atom_mol.long()
        w = edm.loss_weighting(sigma[batch])
        coordinate = (((pos_pred - noise) ** 2).mean(dim=-1) * w).mean()
        losses["coordinate"] = coordinate
        losses["total"] = losses["total"] + loss_weights["coordinate"] * coordinate
        if properties is not None and prop_pred is not None:                                   # model1.py:831-839
            null_rand = draws["null_rand"].to(dev) if "null_rand" in draws else torch.rand(B, device=dev)
            masked = properties.to(dev).float().clone()
            masked[null_rand < 0.1] = 0.0
            property_loss = F.mse_loss(prop_pred, masked)
            losses["property"] = property_loss
            losses["total"] = losses["total"] + loss_weights["property"] * property_loss
        if cons is not None:                                                                   # model1.py:842-886
            # Each term sits in its own try / except in the reference: a label whose shape does not fit is skipped,
            # not fatal (only shape / type errors are swallowed here; anything else still raises).  The consistency
            # outputs carry no gradient in this build, so these terms add their VALUE to losses['total'] (as the
            # reference reports it) but optimising that total does not train the consistency module through them:
            # said once per process when a label is present.
            atom_logits, _, valency_pred, adjacency = cons
            skip = (RuntimeError, ValueError, IndexError, TypeError)
            used = False
            if atom_logits is not None and hasattr(data, "true_atom_types"):
                try:
                    al = F.cross_entropy(atom_logits, data.true_atom_types.to(dev))
                    acc = (atom_logits.argmax(-1) == data.true_atom_types.to(dev)).float().mean().item()
                    losses["atom_type"] = al
                    losses["total"] = losses["total"] + loss_weights["atom_type"] * al
                    metrics["atom_accuracy"] = acc
                    used = True
                except skip:
                    pass
            if valency_pred is not None and hasattr(data, "valency_labels"):
                try:
                    lab = data.valency_labels.to(dev).float()
                    vl = F.mse_loss(valency_pred, lab)
                    mae = F.l1_loss(valency_pred, lab).item()
                    losses["valency"] = vl
                    losses["total"] = losses["total"] + loss_weights["valency"] * vl
                    metrics["valency_mae"] = mae
                    used = True
                except skip:
                    pass
            if adjacency is not None and hasattr(data, "adjacency_matrix"):
                try:
                    pa = adjacency.reshape(-1, 5)
                    ta = data.adjacency_matrix.to(dev).reshape(-1, 5).argmax(-1)
                    adl = F.cross_entropy(pa, ta)
                    acc = (pa.argmax(-1) == ta).float().mean().item()
                    losses["adjacency"] = adl
                    losses["total"] = losses["total"] + loss_weights["adjacency"] * adl
                    metrics["adjacency_accuracy"] = acc
                    used = True
                except skip:
                    pass
            if used and not getattr(ResearchAccurateDiffusionModel, "_warned_detached_consistency", False):
                import warnings
                ResearchAccurateDiffusionModel._warned_detached_consistency = True
                warnings.warn("compute_loss: the atom_type / valency / adjacency terms are computed from detached "
                              "consistency outputs: they are part of losses['total'] as a value, but its backward does "
                              "not reach the consistency module (model1.py:842-886 backpropagates through them)")
        return losses, metrics

    def sample(self, num_samples=1, target_properties=None, guidance_scale=1.0, num_steps=100, max_atoms=30,
               seed=None):
        """model1.py:890-978 without the RDKit post-processing (model1.py:980-1031, host code that
        is not rebuilt): all ``num_samples`` molecules share each forward.  Returns one dict per
        molecule with what the reference passes on to RDKit: 'positions' [k,3] and 'atom_types' [k]
        for the first k = min(15, max_atoms) atoms (model1.py:991-997 clamps types to 1..8 and turns
        a hydrogen that is not atom 0 into carbon), 'num_atoms' = k; there is no 'smiles' key."""
        from .alt_samplers import ddim_positions
        self.eval()
        if seed is None:   # a fresh Philox key per call, drawn from torch's global generator (torch.manual_seed repeats it)
            seed = int(torch.randint(0, 1 << 62, (1,), dtype=torch.int64).item())
        pos, atom_pred = ddim_positions(self, num_samples, target_properties, guidance_scale, num_steps, max_atoms,
                                        seed)
        k = min(15, max_atoms)
        types = atom_pred.argmax(dim=-1).clamp(1, 8).view(num_samples, max_atoms)[:, :k].clone()
        types[:, 1:][types[:, 1:] == 1] = 6
        pos = pos.view(num_samples, max_atoms, 3)[:, :k].cpu().numpy()
        types = types.cpu().numpy()
        return [{"positions": pos[i], "atom_types": types[i], "num_atoms": k} for i in range(num_samples)]


ResearchValidatedDiffusionModel = ResearchAccurateDiffusionModel  # model1.py:1035
