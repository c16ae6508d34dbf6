"""In-situ duration of the attention launch (k_attention_reg) on ragged template ligands.
Usage: python tools/att_bench.py [lo] [hi] [atoms]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_model_ligandmol_generation_amd import ResearchAccurateDiffusionModel
from diffusion_model_ligandmol_generation_amd.engine import template_graph

lo = int(sys.argv[1]) if len(sys.argv) > 1 else 10
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cap = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
g = torch.Generator().manual_seed(555)
sizes, tot = [], 0
for n in torch.randint(lo, hi + 1, (20000,), generator=g).tolist():
    if tot + n > cap:
        break
    sizes.append(n)
    tot += n
dev = torch.device("cuda")
torch.manual_seed(0)
model = ResearchAccurateDiffusionModel(max_atoms=128).to(dev).eval()
eng = model.engine()
graph, _, _ = template_graph(sizes, dev)
x = torch.randn(graph.n_atoms, 119, device=dev)
pos = torch.randn(graph.n_atoms, 3, device=dev)
t = torch.tensor([500], dtype=torch.int32, device=dev)
for mode in (0, 1):
    o = eng.options(gemm_mode=mode, chain_mode=0)
    eng.time_launch("dit1.attention", graph, x, pos, t, reps=3, options=o)
    print(f"{len(sizes)} molecules {lo}-{hi} atoms, N={graph.n_atoms}, gemm_mode={mode}: dit1.attention "
          f"{eng.time_launch('dit1.attention', graph, x, pos, t, reps=10, options=o) * 1e3:.1f} us")
