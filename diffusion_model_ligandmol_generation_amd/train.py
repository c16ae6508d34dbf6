"""Drop-in trainer: the reference's ResearchAccurateTrainer call surface (train.py:27-784)
with the noise / loss step, the score network and its backward pass in the HIP library.

Data parallelism (absent from the reference) is one process per GPU: the global batch is
sharded over ranks, every mean of the loss is taken over the GLOBAL atom / molecule
counts, the first-two-molecules distance term (train.py:642) is evaluated on the ranks
that hold those molecules, and parameter gradients are summed with one all-reduce (RCCL
over xGMI when the backend is "nccl") before clipping, so a sharded step equals the
single-process step on the concatenated batch.
"""
import ctypes as C
import math
import os
import time
from collections import defaultdict

import torch

from . import _lib
from ._lib import check
from .autograd import packed_graph_of
from .backward import EDMLossFunction, _Ctx
from .engine import _ptr, _stream

LOSS_TERMS = ("edm_coordinate", "edm_preconditioning", "pilot_property", "dit_attention", "dit_adaptive_norm",
              "e3_coordinate", "e3_rotation", "e3_translation")


class ResearchAccurateTrainer:
    def __init__(self, model, train_loader, val_loader, test_loader, chemical_constants=None, lr=2e-4,
                 weight_decay=1e-5, debug_mode=False, process_group=None, device=None, reference_collate_quirk=False):
        if device is None:
            device = next(model.parameters()).device
            if device.type != "cuda":
                if not torch.cuda.is_available():
                    raise RuntimeError("ResearchAccurateTrainer needs a HIP device; there is no CPU path")
                device = torch.device("cuda", torch.cuda.current_device())
        self.device = device
        self.model = model.to(device)
        self.train_loader, self.val_loader, self.test_loader = train_loader, val_loader, test_loader
        self.debug_mode = debug_mode
        # True: take ``batch.properties`` exactly as the reference's own loaders deliver them (data_utils.py:904-914
        # stores 14 values per molecule as a 1-D vector, PyG collate concatenates them to [14 B]) and do with them
        # what train.py:327-343 / :388-397 does: one row for every molecule, PILOT loss family skipped.  False: such
        # a tensor is refused, because that behaviour is almost certainly not what a caller wants.
        self.reference_collate_quirk = reference_collate_quirk
        from .optim import FusedAdamW      # a torch.optim.AdamW whose clip + step run in the HIP library when they can
        self.optimizer = FusedAdamW(self.model.parameters(), lr=lr, weight_decay=weight_decay,
                                    betas=(0.9, 0.999), eps=1e-8)                             # train.py:47-53
        self.warmup_steps = 1000
        self.scheduler = torch.optim.lr_scheduler.LambdaLR(self.optimizer, lr_lambda=self._edm_lr_schedule)
        self.chemical_constants = chemical_constants
        self.loss_weights = {                                                                 # train.py:66-90
            "edm_coordinate": 1.0, "edm_preconditioning": 0.1, "pilot_property": 0.2, "pilot_guidance": 0.15,
            "pilot_null_conditioning": 0.05, "moldiff_adjacency": 0.8, "moldiff_valency": 0.6,
            "moldiff_atom_type": 0.4, "moldiff_bond_type": 0.3, "dit_attention": 0.1, "dit_adaptive_norm": 0.05,
            "e3_coordinate": 1.0, "e3_rotation": 0.1, "e3_translation": 0.05}
        self.training_metrics = defaultdict(list)
        self.step_count = 0
        self.research_compliance = {k: True for k in ("edm_preconditioning", "pilot_classifier_free",
                                                      "moldiff_consistency", "graph_dit_attention", "e3_equivariance")}
        self.group = process_group
        self.world = 1
        self.rank = 0
        self.count_group = None
        self.dp_seed_base = 0
        if process_group is not None:
            import torch.distributed as dist
            self.world, self.rank = dist.get_world_size(process_group), dist.get_rank(process_group)
            # the score network's backward sums its parameter gradients over this group (one collective
            # on the flat gradient buffer it already owns)
            self.model.grad_allreduce_group = process_group
            # Per-step bookkeeping of the sharded batch (each rank's atom / molecule count) is exchanged between the
            # HOST processes: with an RCCL main group a gloo group over the same ranks carries it, so no step reads a
            # device tensor back before its forward is enqueued (a device all-gather + .cpu() would drain the GPU).
            # The dropout key of step k is base + k with rank 0's base, broadcast once here.
            self.count_group = process_group
            if dist.get_backend(process_group) == "nccl":
                self.count_group = dist.new_group(ranks=dist.get_process_group_ranks(process_group), backend="gloo")
            base = torch.randint(0, 1 << 61, (1,), dtype=torch.int64)
            dist.broadcast(base, src=dist.get_global_rank(process_group, 0), group=self.count_group)
            self.dp_seed_base = int(base.item())

    def _edm_lr_schedule(self, step):
        """train.py:110-117."""
        if step < self.warmup_steps:
            return step / self.warmup_steps
        progress = (step - self.warmup_steps) / max(1, 50000 - self.warmup_steps)
        return 0.5 * (1 + math.cos(math.pi * min(progress, 1.0)))

    # ------------------------------------------------------------------ loss step
    def _compute_research_exact_loss(self, batch, training=True, draws=None):
        """train.py:267-452.  Returns the reference's dict: total_loss (differentiable), the
        five family totals and the per-term values.  ``draws`` optionally injects the random
        numbers: dict with 'u' [B] uniform (sigma), 'noise' [N,3] normal, 'null_rand' [B]
        uniform (null-conditioning mask)."""
        if batch is None:
            return {"total_loss": torch.tensor(0.0, device=self.device)}
        dev = self.device
        model = self.model
        eng = model.engine()
        lib = eng.lib
        x = batch.x.to(dev).float().contiguous()
        pos = batch.pos.to(dev).float().contiguous()
        N = x.shape[0]
        if getattr(batch, "batch", None) is None:                                   # train.py:281-284
            batch.batch = torch.zeros(N, dtype=torch.long, device=dev)
        batch.batch = batch.batch.to(dev)
        batch.edge_index = batch.edge_index.to(dev)
        graph = packed_graph_of(batch)
        B = graph.n_mols
        draws = draws or {}
        u = draws["u"].to(dev).float() if "u" in draws else torch.rand(B, device=dev)        # train.py:289-293
        noise = draws["noise"].to(dev).float().contiguous() if "noise" in draws else torch.randn_like(pos)
        edm = model.edm_preconditioning
        coef = torch.empty(B, 4, device=dev)
        c_noise = torch.empty(B, device=dev)
        noisy = torch.empty_like(pos)
        pre_pos = torch.empty_like(pos)
        with torch.cuda.device(dev):
            check(lib.ligdiff_edm_prepare(_ptr(u), _ptr(pos), _ptr(noise), _ptr(graph.atom_mol), N, B,
                                          float(edm.sigma_min), float(edm.sigma_max), float(edm.sigma_data),
                                          _ptr(coef), _ptr(c_noise), _ptr(noisy), _ptr(pre_pos), _stream()),
                  "ligdiff_edm_prepare")
        properties = None
        pilot_loss = True
        if hasattr(batch, "properties") and batch.properties is not None:              # train.py:327-343
            properties = batch.properties.to(dev).float()
            if properties.dim() == 1:
                properties = properties.unsqueeze(0)
            if properties.shape[0] != B:
                properties = properties.repeat(B, 1) if properties.shape[0] == 1 else properties[:B]
            if properties.shape[1] != 15:
                if not self.reference_collate_quirk:
                    raise ValueError("batch.properties must be [B,15]; the reference's own loaders deliver [14 B] "
                                     "(data_utils.py:904-914 + PyG collate), which train.py:327-343 turns into one "
                                     "row for every molecule with the PILOT loss skipped: pass "
                                     "reference_collate_quirk=True to the trainer to get exactly that")
                if properties.shape[1] < 8:
                    raise ValueError("batch.properties needs at least the 8 columns the PILOT encoders read")
                # F.mse_loss(prop_pred [B,15], properties [B,w]) raises inside the try block of train.py:388-397:
                # the family contributes nothing; the model reads columns 0..7 only (model1.py:342-345)
                pilot_loss = False
                p15 = properties.new_zeros(B, 15)
                p15[:, :8] = properties[:, :8]
                properties = p15
            if training:
                null_rand = draws["null_rand"].to(dev) if "null_rand" in draws else torch.rand(B, device=dev)
                properties = properties.clone()
                properties[null_rand < 0.1] = 0.0
            properties = properties.contiguous()

        # global counts for data-parallel means, and this shard's position in the global batch
        n_atoms_g, n_mols_g, mol_off, atom_off = N, B, 0, 0
        if self.group is not None:
            import torch.distributed as dist
            cnt = torch.tensor([N, B], dtype=torch.int64)             # host tensors, host-side collective
            allc = [torch.zeros_like(cnt) for _ in range(self.world)]
            dist.all_gather(allc, cnt, group=self.count_group)
            allc = torch.stack(allc)
            n_atoms_g, n_mols_g = int(allc[:, 0].sum()), int(allc[:, 1].sum())
            mol_off, atom_off = int(allc[:self.rank, 1].sum()), int(allc[:self.rank, 0].sum())
            shared_seed = (self.dp_seed_base + self.step_count) & ((1 << 62) - 1)   # the same key on every rank

        class _Data:
            pass

        d = _Data()
        d.x, d.pos, d.edge_index, d.batch, d._ligdiff_graph = x, pre_pos, batch.edge_index, batch.batch, graph
        # dropout masks are keyed by (seed, global atom index): with the same seed on every rank the sharded
        # step draws the mask of the unsharded batch
        d.atom_offset = atom_off
        if "dropout_seed" in draws:
            d.dropout_seed = int(draws["dropout_seed"])
        elif self.group is not None:
            d.dropout_seed = shared_seed
        # the consistency module's outputs reach no loss term here (train.py:519, :543, :559, :583), so the
        # training step does not spend an O(sum n^2) pass and an [N,N] tensor on them
        d.skip_consistency = not getattr(self, "run_consistency", False)
        atom_pred, pos_pred, prop_pred, _ = model(d, c_noise, properties)               # train.py:349

        rot_mols = max(0, min(2, n_mols_g) - mol_off)
        rot_mols = min(rot_mols, B)
        loss_props = properties if pilot_loss else None      # what the PILOT loss family sees
        st = _Ctx(eng=eng, graph=graph, pos=pos, noise=noise, noisy=noisy, coef=coef, props=loss_props,
                  rot_mols=rot_mols, n_atoms_global=n_atoms_g, n_mols_global=n_mols_g, group=self.group)
        out = EDMLossFunction.apply(st, atom_pred, pos_pred, prop_pred if loss_props is not None else None)
        properties = loss_props
        losses = {"total_loss": out[8]}
        vals = out.detach()
        for i, k in enumerate(LOSS_TERMS):
            if k == "pilot_property" and properties is None:
                continue
            losses[k] = vals[i]
        zero = torch.tensor(0.0, device=dev)
        if properties is not None:
            losses["pilot_guidance"] = zero
            losses["pilot_null_conditioning"] = zero
        losses["edm_total"] = vals[0] + vals[1]
        losses["pilot_total"] = vals[2] if properties is not None else zero
        losses["moldiff_total"] = zero
        losses["dit_total"] = vals[3] + vals[4]
        losses["e3_total"] = vals[5] + vals[6] + vals[7]
        return losses

    # ------------------------------------------------------------------ optimisation step
    def _allreduce_gradients(self):
        """Gradient sum over the ranks.  Normally already done: the score network's backward reduces
        its flat gradient buffer in place (``model.grad_allreduce_group``, one collective, no copies).
        If that hook was removed, fall back to one flat all-reduce over every gradient that exists
        (the same set on every rank: it depends only on whether properties are present)."""
        if self.group is None or getattr(self.model, "grad_allreduce_group", None) is not None:
            return
        from .parallel import allreduce_flat
        allreduce_flat([p.grad for p in self.model.parameters() if p.grad is not None], self.group)

    def train_step(self, batch, draws=None):
        """train.py:187-200 for one batch (one shard of the global batch under data parallelism)."""
        self.optimizer.zero_grad()
        loss_dict = self._compute_research_exact_loss(batch, training=True, draws=draws)
        loss_dict["total_loss"].backward()
        self._allreduce_gradients()
        if hasattr(self.optimizer, "clip_and_step"):      # train.py:196-197 as two launches over the flat gradient buffer
            self.optimizer.clip_and_step(self.model, max_norm=1.0)
        else:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), max_norm=1.0)
            self.optimizer.step()
        self.step_count += 1
        return loss_dict

    def _train_epoch_research_exact(self):
        """train.py:176-232."""
        self.model.train()
        epoch_metrics = defaultdict(float)
        num_batches = 0
        for batch in self.train_loader:
            loss_dict = self.train_step(batch)
            for k, v in loss_dict.items():
                epoch_metrics[k] += float(v)
            num_batches += 1
        return {k: v / max(1, num_batches) for k, v in epoch_metrics.items()}

    def _validate_epoch_research_exact(self):
        """train.py:234-265."""
        self.model.eval()
        val_metrics = defaultdict(float)
        num_batches = 0
        with torch.no_grad():
            for batch in self.val_loader:
                loss_dict = self._compute_research_exact_loss(batch, training=False)
                for k, v in loss_dict.items():
                    val_metrics[k] += float(v)
                num_batches += 1
        return {k: v / max(1, num_batches) for k, v in val_metrics.items()}

    def train_research_validated(self, num_epochs=100, validation_frequency=5, save_frequency=10):
        """train.py:119-174: epochs, periodic validation with early stopping (patience 15),
        periodic checkpoints, one scheduler step per EPOCH (as the reference does)."""
        best_val_loss, patience, max_patience = float("inf"), 0, 15
        for epoch in range(num_epochs):
            t0 = time.time()
            tm = self._train_epoch_research_exact()
            tm["epoch_time"] = time.time() - t0
            for k, v in tm.items():
                self.training_metrics["train_" + k].append(v)
            if (epoch + 1) % validation_frequency == 0 and self.val_loader is not None:
                vm = self._validate_epoch_research_exact()
                for k, v in vm.items():
                    self.training_metrics["val_" + k].append(v)
                if vm.get("total_loss", float("inf")) < best_val_loss:
                    best_val_loss, patience = vm["total_loss"], 0
                    self._save_research_checkpoint(epoch, vm, is_best=True)
                else:
                    patience += 1
                if patience >= max_patience:
                    break
            if (epoch + 1) % save_frequency == 0:
                self._save_research_checkpoint(epoch, tm, is_best=False)
            self.scheduler.step()
        return self.training_metrics

    def _validate_research_compliance(self):
        """train.py:693-734: the five sub-modules exist."""
        m = self.model
        return all(hasattr(m, a) for a in ("edm_preconditioning", "property_guidance", "moldiff_consistency",
                                           "dit_layers", "equivariant_layers"))

    def _save_research_checkpoint(self, epoch, metrics, is_best=False):
        """train.py:736-784: the same dictionary (keys, nested configuration blocks) under the same
        file names; written by rank 0 only under data parallelism."""
        if self.rank != 0:
            return
        os.makedirs("research_checkpoints", exist_ok=True)
        m = self.model
        ckpt = {"epoch": epoch, "step": self.step_count, "model_state_dict": m.state_dict(),
                "optimizer_state_dict": self.optimizer.state_dict(),
                "scheduler_state_dict": self.scheduler.state_dict(), "metrics": metrics,
                "training_metrics": dict(self.training_metrics),
                "research_compliance": self.research_compliance, "loss_weights": self.loss_weights,
                "model_config": {"timesteps": getattr(m, "timesteps", 1000), "hidden_dim": getattr(m, "hidden_dim", 256),
                                 "atom_feature_dim": getattr(m, "atom_feature_dim", 119),
                                 "max_atoms": getattr(m, "max_atoms", 100)},
                "research_specs": {"edm_sigma_min": 0.002, "edm_sigma_max": 80.0, "edm_sigma_data": 0.5,
                                   "pilot_objectives": 8, "moldiff_bond_types": 5, "dit_num_heads": 8,
                                   "e3_equivariance": True}}
        if is_best:
            torch.save(ckpt, os.path.join("research_checkpoints", "best_research_model.pt"))
        torch.save(ckpt, os.path.join("research_checkpoints", f"checkpoint_epoch_{epoch}.pt"))
        torch.save(ckpt, os.path.join("research_checkpoints", "latest_checkpoint.pt"))

    def generate_research_validated_molecules(self, num_samples=10, target_properties=None, guidance_scale=7.5,
                                              num_steps=50, temperature=1.0, top_k=None, seed=None):
        """train.py:786-892 without the RDKit post-processing (train.py:916-1036, host code with
        random atom typing that is not rebuilt): 25-atom chain + ring graphs, EDM sigma schedule,
        Euler coordinate updates, all molecules in one batch.  Returns one dict per molecule with
        what the reference hands to RDKit: 'positions' [15,3] (train.py:927) and 'num_atoms'."""
        from .alt_samplers import euler_positions
        self.model.eval()
        if seed is None:   # a fresh Philox key per call, drawn from torch's global generator
            seed = int(torch.randint(0, 1 << 62, (1,), dtype=torch.int64).item())
        pos = euler_positions(self.model, num_samples, target_properties, guidance_scale, num_steps, temperature,
                              25, seed)
        pos = pos.view(num_samples, 25, 3)[:, :15].cpu().numpy()
        return [{"positions": pos[i], "num_atoms": 15} for i in range(num_samples)]


ResearchValidatedTrainer = ResearchAccurateTrainer
