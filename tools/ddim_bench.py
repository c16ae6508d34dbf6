"""Wall time of one batched DDIM call (the reference generator's default: 100 steps) on 1024 x 29-atom
template ligands, graph replay against the plain launch loop.  Usage: python tools/ddim_bench.py [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_model_ligandmol_generation_amd import ResearchAccurateDiffusionModel
from diffusion_model_ligandmol_generation_amd.generation import ResearchValidatedGenerator

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
torch.manual_seed(0)
model = ResearchAccurateDiffusionModel().cuda().eval()
gen = ResearchValidatedGenerator(model)
sizes = [29] * 1024
for use_graph in (True, False):
    for _ in range(2):
        gen.sample_batch(sizes, None, 1.0, 1.0, use_ddim=True, num_steps=steps, seed=1, use_graph=use_graph)
    torch.cuda.synchronize()
    t0 = time.time()
    reps = 5
    for _ in range(reps):
        gen.sample_batch(sizes, None, 1.0, 1.0, use_ddim=True, num_steps=steps, seed=1, use_graph=use_graph)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / reps
    print(f"DDIM {steps} steps, use_graph={use_graph}: {dt*1e3:.1f} ms per call = {dt/steps*1e3:.3f} ms per step, {len(sizes)/dt:.0f} molecules/s")
