"""Per-pass wall time of the headline workload (1024 x 29-atom ligands, 1000-step DDPM) in one process:
shows whether the first process on a machine, or the first passes of a process, run slower than the rest.
Usage: python tools/cold_start.py [passes] [one-call]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_model_ligandmol_generation_amd import ResearchAccurateDiffusionModel
from diffusion_model_ligandmol_generation_amd.generation import ResearchValidatedGenerator

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 6
torch.manual_seed(0)
model = ResearchAccurateDiffusionModel().cuda().eval()
gen = ResearchValidatedGenerator(model)
if len(sys.argv) > 2 and sys.argv[2] == "one-call":
    gen.concurrent_atoms = None
sizes = [29] * 1024
for i in range(passes):
    torch.cuda.synchronize()
    t0 = time.time()
    with torch.no_grad():
        gen.sample_batch(sizes, None, 1.0, 1.0, seed=2024, atom_offset=i * 29696)
    torch.cuda.synchronize()
    print(f"pass {i}: {time.time() - t0:.3f} s", flush=True)
