"""End-to-end sanity beyond random weights: a few hundred training steps of the HIP trainer on seeded synthetic
QM9-size ligands (the trainer's own epoch loop, train.py:119-232: one scheduler step per epoch), then a batched
1000-step DDPM sampling run through the generator's outer loop with its validity gates (generation.py:162-233),
reporting the loss curve, the gates' acceptance rate and the share of distinct molecules among the samples.

Two settings differ from the reference's defaults so that a run of seconds moves the weights at all, and are
reported in the output: lr 1e-3 instead of 1e-4 and a 10-epoch instead of a 1000-epoch warm-up (train.py:110-117
counts epochs).  Distinctness is judged on a fingerprint that needs no RDKit: the multiset of predicted atom types
(argmax of the 119 features) together with the sorted pairwise distances rounded to 0.1.

Usage: python tools/train_and_sample.py [epochs] [molecules_to_sample]   -> one JSON object on stdout."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def fingerprint(x, pos):
    types = tuple(sorted(np.argmax(x, axis=1).tolist()))
    d = np.sqrt(((pos[:, None, :] - pos[None, :, :]) ** 2).sum(-1))
    iu = np.triu_indices(len(pos), 1)
    return types, tuple(np.round(np.sort(d[iu]), 1).tolist())


def run(epochs=40, n_sample=256, n_train=512, batch_size=64, device="cuda"):
    from run_config1 import synthetic_ligands
    from diffusion_model_ligandmol_generation_amd import ResearchAccurateDiffusionModel
    from diffusion_model_ligandmol_generation_amd.data import MoleculeLoader
    from diffusion_model_ligandmol_generation_amd.generation import ResearchValidatedGenerator
    from diffusion_model_ligandmol_generation_amd.train import ResearchAccurateTrainer
    mols = synthetic_ligands(n_train + 64, seed=2025)
    torch.manual_seed(0)
    np.random.seed(0)
    train_loader = MoleculeLoader(mols[:n_train], batch_size=batch_size, shuffle=True, drop_last=True)
    val_loader = MoleculeLoader(mols[n_train:], batch_size=batch_size)
    model = ResearchAccurateDiffusionModel().to(device)
    trainer = ResearchAccurateTrainer(model, train_loader, val_loader, None, lr=1e-3, weight_decay=1e-6)
    trainer.warmup_steps = 10
    trainer.scheduler = torch.optim.lr_scheduler.LambdaLR(trainer.optimizer, lr_lambda=trainer._edm_lr_schedule)
    gen = ResearchValidatedGenerator(model)
    np.random.seed(1)
    sizes = [gen._sample_realistic_molecule_size((5, 29)) for _ in range(n_sample)]

    def sample(seed):
        model.eval()
        torch.cuda.synchronize()
        t0 = time.time()
        results, flags = gen.sample_many(sizes, None, 1.0, 1.0, use_ddim=False, seed=seed)
        torch.cuda.synchronize()
        dt = time.time() - t0
        finite = [bool(np.isfinite(r[0]).all() and np.isfinite(r[1]).all()) for r in results]
        prints = {fingerprint(r[0], r[1]) for r, ok in zip(results, finite) if ok}
        return {"molecules": len(sizes), "seconds": dt, "accepted_by_gates": int(sum(flags)),
                "acceptance_rate": float(sum(flags)) / len(sizes), "finite": int(sum(finite)),
                "distinct_among_finite": len(prints) / max(1, sum(finite))}

    out = {"workload": f"{n_train} synthetic 5-29-atom ligands, batch {batch_size}, {epochs} epochs of the HIP trainer "
                       f"(train mode, dropout 0.1), then {n_sample} molecules through a 1000-step DDPM and the validity gates",
           "settings": {"lr": 1e-3, "warmup_epochs": 10, "reference_defaults": {"lr": 1e-4, "warmup_epochs": 1000}}}
    out["before_training"] = sample(11)
    curve, t0 = [], time.time()
    for epoch in range(epochs):
        tm = trainer._train_epoch_research_exact()
        trainer.scheduler.step()
        curve.append(tm["total_loss"])
    torch.cuda.synchronize()
    dt = time.time() - t0
    vm = trainer._validate_epoch_research_exact()
    out["training"] = {"steps": trainer.step_count, "seconds": dt, "ms_per_step": dt / max(1, trainer.step_count) * 1e3,
                       "train_total_loss_by_epoch": [round(float(v), 4) for v in curve],
                       "validation_total_loss": float(vm["total_loss"]),
                       "all_finite": bool(np.isfinite(curve).all())}
    out["after_training"] = sample(11)
    return out


if __name__ == "__main__":
    e = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    print(json.dumps(run(e, n)))
