"""The 100k-molecule sampling sweep of the mixed-size ragged workload, run for real on one GPU: 10-128-atom
template ligands (seeded uniform sizes), 1000-step DDPM, MLP GEMMs in split-operand mode, packed batches of the
generator's own size (131 072 atoms in split-operand mode, 32 768 on the fp32 pipe), results copied to the host and
judged there (finite, gates, distinct fingerprints).  The request is handed to the generator 16 000 molecules at a time.

Usage: python tools/config5_sweep.py [molecules] [out.json] [split(1|0)] [atoms per packed batch]
Prints a progress line per 16 000 molecules on stderr and one JSON object on stdout (and into out.json)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def fingerprint(x, pos):
    """Multiset of argmax atom types + sorted pairwise distances rounded to 0.1 (no RDKit needed)."""
    types = tuple(sorted(np.argmax(x, axis=1).tolist()))
    d = np.sqrt(((pos[:, None, :] - pos[None, :, :]) ** 2).sum(-1))
    iu = np.triu_indices(len(pos), 1)
    return hash((types, tuple(np.round(np.sort(d[iu]), 1).tolist())))


def run(n_mols=100000, split=True, chunk=16000, device="cuda", log=sys.stderr, batch_atoms=None):
    from diffusion_model_ligandmol_generation_amd import ResearchAccurateDiffusionModel
    from diffusion_model_ligandmol_generation_amd.generation import ResearchValidatedGenerator
    g = torch.Generator().manual_seed(555)
    sizes = torch.randint(10, 129, (n_mols,), generator=g).tolist()
    torch.manual_seed(0)
    model = ResearchAccurateDiffusionModel(max_atoms=128).to(device).eval()
    gen = ResearchValidatedGenerator(model)
    eng = model.engine()
    gen.options = eng.options(gemm_mode=int(bool(split)))      # per call; the process-wide mode is left alone
    if batch_atoms:
        gen.batch_atoms = gen.batch_atoms_split = int(batch_atoms)
    finite = accepted = 0
    prints = set()
    gpu_s = 0.0
    t_all = time.time()
    if True:
        offset = 0
        for lo in range(0, n_mols, chunk):
            part = sizes[lo:lo + chunk]
            torch.cuda.synchronize()
            t0 = time.time()
            res, flags = gen.sample_many(part, None, 1.0, 1.0, use_ddim=False, seed=31, atom_offset=offset)
            torch.cuda.synchronize()
            gpu_s += time.time() - t0
            offset += sum(part)
            accepted += int(sum(flags))
            for x, pos in res:
                if np.isfinite(x).all() and np.isfinite(pos).all():
                    finite += 1
                    prints.add(fingerprint(x, pos))
            done = lo + len(part)
            print(f"[config5 sweep] {done}/{n_mols} molecules, {gpu_s:.1f} s sampling, {done / gpu_s:.1f} molecules/s, "
                  f"{time.time() - t_all:.1f} s wall", file=log, flush=True)
    return {"workload": "ragged 10-128-atom template ligands (uniform sizes, seed 555), 1000-step DDPM, "
                        + ("split-operand bf16 MFMA GEMMs (fp32 accumulate)" if split else "fp32 matrix pipe")
                        + f", batches of <= {gen.batch_atoms_split if split else gen.batch_atoms} atoms, results copied to the host",
            "molecules": n_mols, "atoms": int(sum(sizes)), "ddpm_steps": 1000,
            "sampling_seconds": gpu_s, "wall_seconds_with_host_checks": time.time() - t_all,
            "molecules_per_s": n_mols / gpu_s, "atom_steps_per_s": sum(sizes) * 1000.0 / gpu_s,
            "finite": finite, "accepted_by_gates": accepted, "acceptance_rate": accepted / n_mols,
            "distinct_among_finite": len(prints) / max(1, finite)}


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    path = sys.argv[2] if len(sys.argv) > 2 else None
    sp = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
    ba = int(sys.argv[4]) if len(sys.argv) > 4 else None
    out = run(n, sp, batch_atoms=ba)
    line = json.dumps(out)
    if path:
        with open(path, "w") as fh:
            fh.write(line + "\n")
    print(line)
