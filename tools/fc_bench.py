"""Large fully connected ligands (65-128 atoms, the seeded sizes of bench.py's config-4 leg): the fused edge kernel
with and without the in-kernel per-destination sums (edge_reduce), launch by launch and as ms per reverse step of a
short DDPM call.  Usage: python tools/fc_bench.py [molecules] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_model_ligandmol_generation_amd import ResearchAccurateDiffusionModel
from diffusion_model_ligandmol_generation_amd.engine import full_graph
from diffusion_model_ligandmol_generation_amd.generation import ResearchValidatedGenerator

B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
g = torch.Generator().manual_seed(4321)
sizes = torch.randint(65, 129, (B,), generator=g).tolist()
dev = torch.device("cuda")
torch.manual_seed(0)
model = ResearchAccurateDiffusionModel(max_atoms=128, timesteps=steps).to(dev).eval()
eng = model.engine()
gen = ResearchValidatedGenerator(model)
graph, _ = full_graph(sizes, dev)
x = torch.randn(graph.n_atoms, 119, device=dev)
pos = torch.randn(graph.n_atoms, 3, device=dev)
t = torch.tensor([steps // 2], dtype=torch.int32, device=dev)
print(f"N={graph.n_atoms} E={graph.n_edges}")
for red in (0, 1):
    o = eng.plan_graph(graph, eng.options(edge_reduce=red))
    line = [f"edge_reduce={red} plan={o.as_dict()}"]
    for lab in ("e3_1.edge_geom", "e3_1.edge_chain", "e3_1.aggregate", "e3_1.node_mlp1", "e3_1.node_mlp2"):
        try:
            eng.time_launch(lab, graph, x, pos, t, reps=2, options=o)
            line.append(f"{lab} {eng.time_launch(lab, graph, x, pos, t, reps=5, options=o) * 1e3:.1f} us")
        except Exception as e:   # noqa: BLE001
            line.append(f"{lab} -")
    print("; ".join(line), flush=True)
    gen.options = eng.options(edge_reduce=red)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        with torch.no_grad():
            xs, ps, _, _ = gen.sample_batch(sizes, None, 1.0, 1.0, seed=77, connectivity="full")
        torch.cuda.synchronize()
        dt = time.time() - t0
    print(f"  {steps}-step DDPM call: {dt * 1e3 / steps:.3f} ms per reverse step (finite: {bool(torch.isfinite(xs).all())})", flush=True)
