"""BASELINE configs[0] end to end on the HIP path: the flow of the reference's main.py:187-274
(split -> loaders -> model -> trainer -> one training step -> generation) on seeded synthetic
QM9-size ligands (5-29 atoms, bond templates), batch 64, one train step and a 100-step DDPM, with
the CPU oracle run beside it on the same inputs and draws.

main.py itself cannot run (it imports a module named `model` that the reference does not contain and
passes constructor arguments model1.py does not accept, main.py:236-249); this restates its steps with
the model's real defaults.  Usage: python tools/run_config1.py  -> one JSON object on stdout."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def synthetic_ligands(n_mols, seed=1234, lo=5, hi=29):
    """Seeded QM9-size molecules: one-hot atom types over {H, C, N, O, F}, coordinates N(0, 1.5^2), the
    generator's bond template (generation.py:562-668), 15 property targets in [0, 1)."""
    from diffusion_model_ligandmol_generation_amd.generation import ResearchValidatedGenerator
    g = torch.Generator().manual_seed(seed)
    conn = ResearchValidatedGenerator.create_research_connectivity_fixed
    mols = []
    for _ in range(n_mols):
        n = int(torch.randint(lo, hi + 1, (1,), generator=g))
        types = torch.tensor([1, 6, 7, 8, 9])[torch.multinomial(torch.tensor([.45, .35, .08, .10, .02]), n, True, generator=g)]
        ei = torch.tensor(conn(None, n)[0], dtype=torch.int64).t().contiguous()
        mols.append({"x": torch.nn.functional.one_hot(types, 119).float(), "pos": torch.randn(n, 3, generator=g) * 1.5,
                     "edge_index": ei, "properties": torch.rand(15, generator=g)})
    return mols


def run(device="cuda", batch_size=64, n_mols=80, ddpm_steps=100, cpu_sample_mols=4):
    import oracle
    from diffusion_model_ligandmol_generation_amd import ResearchAccurateDiffusionModel
    from diffusion_model_ligandmol_generation_amd.data import MoleculeLoader
    from diffusion_model_ligandmol_generation_amd.generation import ResearchValidatedGenerator
    from diffusion_model_ligandmol_generation_amd.train import ResearchAccurateTrainer
    out = {"workload": f"configs[0]: synthetic QM9-size ligands (5-29 atoms), batch {batch_size}: main.py:187-274 flow (num_layers = 8 as main.py:242), "
                       f"1 train step + {ddpm_steps}-step DDPM, HIP path with the CPU oracle beside it"}
    # ---- step 3: split (main.py:191-202) and loaders (main.py:205-228)
    mols = synthetic_ligands(n_mols)
    train_size, val_size = int(0.8 * len(mols)), int(0.1 * len(mols))
    np.random.seed(42)
    idx = np.random.permutation(len(mols))
    train = [mols[i] for i in idx[:train_size]]
    val = [mols[i] for i in idx[train_size:train_size + val_size]]
    test = [mols[i] for i in idx[train_size + val_size:]]
    torch.manual_seed(0)
    train_loader = MoleculeLoader(train, batch_size=batch_size, shuffle=True, drop_last=True)
    val_loader = MoleculeLoader(val, batch_size=batch_size)
    test_loader = MoleculeLoader(test, batch_size=batch_size)
    out["split"] = {"train": len(train), "val": len(val), "test": len(test), "train_batches": len(train_loader)}
    # ---- step 4: model as main.py:238-249 builds it (num_layers = 8: four E(3) and four attention layers;
    # timesteps = the sampler's step count)
    torch.manual_seed(0)
    model = ResearchAccurateDiffusionModel(atom_feature_dim=119, edge_feature_dim=5, hidden_dim=256, num_layers=8,
                                           num_heads=8, timesteps=ddpm_steps, property_dim=15, use_equivariance=True,
                                           use_consistency=True, use_multi_objective=True).to(device)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    # ---- step 5: trainer, ONE training step (main.py:258-274 runs epochs of these)
    trainer = ResearchAccurateTrainer(model, train_loader, val_loader, test_loader, lr=1e-4, weight_decay=1e-6)
    model.train()
    batch = next(iter(train_loader))
    B, N = int(batch.batch.max()) + 1, batch.x.shape[0]
    g = torch.Generator().manual_seed(7)
    draws = {"u": torch.rand(B, generator=g), "noise": torch.randn(N, 3, generator=g), "null_rand": torch.rand(B, generator=g),
             "dropout_seed": 4242}
    trainer.train_step(batch.to(device), draws=draws)                  # warm-up (first-touch allocations)
    torch.cuda.synchronize()
    t0 = time.time()
    losses = trainer.train_step(batch, draws=draws)
    torch.cuda.synchronize()
    out["train_step"] = {"molecules": B, "atoms": N, "ms": (time.time() - t0) * 1e3,
                         "total_loss": float(losses["total_loss"].detach()), "learning_rate": trainer.optimizer.param_groups[0]["lr"]}
    # the same step in the oracle (same weights: the first epoch's learning rate is 0, train.py:110-117 / :167),
    # with the dropout masks the library drew
    eng = model.engine()
    spec = model.dropout_spec(0, 4242)
    masks = [{"att": eng.dropout_mask(spec, l, 0, N).cpu(), "ffn_hidden": eng.dropout_mask(spec, l, 1, N).cpu(),
              "ffn_out": eng.dropout_mask(spec, l, 2, N).cpu()} for l in range(len(model.dit_layers))]
    cb = batch.to("cpu")
    t0 = time.time()
    pn = {n for n, _ in model.named_parameters()}
    sdg = {k: (v.clone().requires_grad_(True) if k in pn else v) for k, v in sd.items()}
    L, _ = oracle.loss_step(sdg, cb.x, cb.pos, cb.edge_index, cb.batch, draws["u"], draws["noise"], cb.properties,
                            draws["null_rand"], dropout=masks)
    L["total_loss"].backward()
    out["train_step"]["oracle_cpu_ms"] = (time.time() - t0) * 1e3
    out["train_step"]["oracle_total_loss"] = float(L["total_loss"].detach())
    out["train_step"]["rel_loss_diff"] = abs(float(losses["total_loss"].detach()) - float(L["total_loss"].detach())) / abs(float(L["total_loss"].detach()))
    # the trainer's gradients are those left behind by clip_grad_norm_(max_norm=1.0) (train.py:196): clip the oracle's alike
    norm = float(torch.sqrt(sum((v.grad.double() ** 2).sum() for v in sdg.values() if getattr(v, "grad", None) is not None)))
    clip = min(1.0, 1.0 / (norm + 1e-6))
    worst = 0.0
    for name, p in model.named_parameters():
        if p.grad is not None and sdg[name].grad is not None:
            ref = sdg[name].grad * clip
            worst = max(worst, float((p.grad.cpu() - ref).abs().max()) / (float(ref.abs().max()) + 1e-12))
    out["train_step"]["gradient_norm_before_clip"] = norm
    out["train_step"]["worst_rel_gradient_diff"] = worst
    # ---- step 6: generation (100-step DDPM, batch 64, unguided)
    model.eval()
    gen = ResearchValidatedGenerator(model)
    np.random.seed(1)
    sizes = [gen._sample_realistic_molecule_size((5, 29)) for _ in range(batch_size)]
    gen.sample_many(sizes[:4], None, 1.0, 1.0, seed=3)                 # warm-up
    torch.cuda.synchronize()
    t0 = time.time()
    results, flags = gen.sample_many(sizes, None, 1.0, 1.0, use_ddim=False, seed=5)
    torch.cuda.synchronize()
    dt = time.time() - t0
    out["sampling"] = {"molecules": batch_size, "ddpm_steps": ddpm_steps, "seconds": dt, "molecules_per_s": batch_size / dt,
                       "accepted": int(sum(flags)), "finite": bool(all(np.isfinite(r[0]).all() for r in results))}
    # CPU oracle, one molecule at a time as generation.py runs it, with the library's own noise (Philox export)
    sched = oracle.cosine_schedule(ddpm_steps)
    k = min(cpu_sample_mols, batch_size)
    ptr = np.cumsum([0] + sizes)
    drift = 0.0
    t0 = time.time()
    with torch.no_grad():
        for m in range(k):
            a0, n = int(ptr[m]), sizes[m]
            draws_m = []
            draws_m.append(eng.philox_normal(n * 119, 5, 0xFFFFFFFF, 0, a0 * 119).view(n, 119).cpu())
            draws_m.append(eng.philox_normal(n * 3, 5, 0xFFFFFFFF, 1, a0 * 3).view(n, 3).cpu())
            for t in range(ddpm_steps - 1, 1, -1):
                draws_m.append(eng.philox_normal(n * 119, 5, t, 0, a0 * 119).view(n, 119).cpu())
                draws_m.append(eng.philox_normal(n * 3, 5, t, 1, a0 * 3).view(n, 3).cpu())
            it = iter(draws_m)
            x, pos = oracle.ddpm_sample(sd, sched, n, lambda shape: next(it), None, 1.0, 1.0, max_atoms=100, pad=True)
            rx = torch.from_numpy(results[m][0])
            drift = max(drift, float((x - rx).abs().max()) / (float(rx.abs().max()) + 1e-12))
    cpu_dt = time.time() - t0
    out["sampling"]["oracle_cpu"] = {"molecules": k, "seconds": cpu_dt, "molecules_per_s": k / cpu_dt,
                                     "cores": torch.get_num_threads(),
                                     "max_rel_trajectory_drift_vs_hip": drift}
    return out


if __name__ == "__main__":
    print(json.dumps(run()))
