"""Per-launch timing of one score-network forward at the benchmark shape (HIP events inside
the library, one labelled launch at a time).  Usage: python tools/profile_labels.py [mols]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_model_ligandmol_generation_amd import ResearchAccurateDiffusionModel
from diffusion_model_ligandmol_generation_amd.engine import template_graph

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.manual_seed(0)
model = ResearchAccurateDiffusionModel(max_atoms=100).cuda().eval()
eng = model.engine()
g, _, _ = template_graph([29] * B, torch.device("cuda"))
x = torch.randn(g.n_atoms, 119, device="cuda")
pos = torch.randn(g.n_atoms, 3, device="cuda")
t = torch.tensor([500], dtype=torch.int32, device="cuda")
labels = ["time.fourier", "time.mlp0", "time.mlp2", "dit.adaln", "embed"]
for l in range(3):
    labels += [f"e3_{l}.{k}" for k in ("node_proj", "edge_geom", "edge_pre", "edge_mlp2", "coord", "edge_chain", "aggregate", "node_mlp1", "node_mlp2", "node_chain")]
labels += ["dit0.ln_mod"]
for l in range(3):
    labels += [f"dit{l}.{k}" for k in ("qkv", "attention", "out_proj", "ffn_up", "ffn_down", "ffn", "chain")]
labels += ["head.hidden", "head.atom", "head.pos"]
tot = 0.0
for lab in labels:
    try:
        ms = eng.time_launch(lab, g, x, pos, t, reps=20)
    except Exception as e:
        print(f"{lab:18s}  n/a ({e})"); continue
    tot += ms
    print(f"{lab:18s} {ms*1e3:8.1f} us")
print(f"{'sum':18s} {tot*1e3:8.1f} us")
