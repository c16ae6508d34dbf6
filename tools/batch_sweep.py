"""Reverse-step time against batch size (29-atom template ligands, unguided DDPM, graph replay) in both
GEMM modes: how far the benchmark's 1024 molecules per GPU are from filling the part.
Usage: python tools/batch_sweep.py [steps [molecules ...]]
SWEEP_CONCURRENT_ATOMS in the environment overrides the generator's threshold for running a batch as two
concurrent halves ("none": never)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_model_ligandmol_generation_amd import ResearchAccurateDiffusionModel
from diffusion_model_ligandmol_generation_amd.generation import ResearchValidatedGenerator

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
torch.manual_seed(0)
model = ResearchAccurateDiffusionModel(timesteps=steps).cuda().eval()
gen = ResearchValidatedGenerator(model)
eng = model.engine()
if "SWEEP_CONCURRENT_ATOMS" in os.environ:
    v = os.environ["SWEEP_CONCURRENT_ATOMS"]
    gen.concurrent_atoms = None if v == "none" else int(v)
print(f"{'molecules':>9} {'atoms':>7} | {'fp32 ms/step':>12} {'mol/s @1000':>11} | {'split ms/step':>13} {'mol/s @1000':>11}")
batches = [int(a) for a in sys.argv[2:]] or [128, 256, 512, 1024, 1130, 2048, 2260, 4096, 8192]
for B in batches:
    sizes = [29] * B
    row = []
    for split in (False, True):
        eng.set_split_gemm(split)
        gen.sample_batch(sizes, None, 1.0, 1.0, seed=1)
        torch.cuda.synchronize()
        t0 = time.time()
        reps = 3
        for _ in range(reps):
            gen.sample_batch(sizes, None, 1.0, 1.0, seed=1)
        torch.cuda.synchronize()
        ms = (time.time() - t0) / reps / steps * 1e3
        row += [ms, B / (ms * 1e-3 * 1000)]
    eng.set_split_gemm(False)
    print(f"{B:9d} {B*29:7d} | {row[0]:12.3f} {row[1]:11.0f} | {row[2]:13.3f} {row[3]:11.0f}")
