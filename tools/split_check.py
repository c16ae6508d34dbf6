"""A/B of the forward pass between the fp32 matrix pipe and the split-bf16 mode: timing, output
difference and per-launch times on the benchmark batch (1024 x 29 atoms)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_model_ligandmol_generation_amd import ResearchAccurateDiffusionModel
from diffusion_model_ligandmol_generation_amd.engine import PackedGraph, template_graph

torch.manual_seed(0)
dev = torch.device("cuda")
model = ResearchAccurateDiffusionModel().to(dev).eval()
eng = model.engine()
B, n = 1024, 29
g = template_graph([n] * B, dev)[0]
N = B * n
x = torch.randn(N, 119, device=dev)
pos = torch.randn(N, 3, device=dev)
t = torch.tensor([500], dtype=torch.int32, device=dev)
outs = {}
labels = ["embed"] + [f"e3_1.{k}" for k in ("node_proj", "edge_pre", "edge_mlp2", "coord", "node_mlp1", "node_mlp2")] \
    + [f"dit1.{k}" for k in ("qkv", "out_proj", "ffn_up", "ffn_down")] + ["head.hidden"]
times = {}
for mode in (False, True):
    eng.set_split_gemm(mode)
    for _ in range(2):
        r = eng.forward(g, x, pos, None, None, want_props=False, t_index=t)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(10):
        eng.forward(g, x, pos, None, None, want_props=False, t_index=t)
    torch.cuda.synchronize()
    print("split" if mode else "fp32 ", "forward ms", (time.time() - t0) * 100)
    outs[mode] = [o.clone() for o in r[:2]]
    times[mode] = {}
    for lab in labels:
        try:
            times[mode][lab] = eng.time_launch(lab, g, x, pos, t, reps=5) * 1e3
        except Exception as e:
            times[mode][lab] = float("nan")
for i, name in enumerate(["atom", "pos"]):
    d = (outs[True][i] - outs[False][i]).abs().max().item()
    s = outs[False][i].abs().max().item()
    print(name, "max abs diff", d, "scale", s)
for lab in labels:
    print(f"{lab:18s} fp32 {times[False][lab]:8.1f} us   split {times[True][lab]:8.1f} us")
