"""Forward error against the CPU oracle on the ragged / fully connected test batches, for A/B runs of the
fused chain kernels (LIGDIFF_CHAIN=force / 0 selects the fused kernels / the separate launches).  Usage: python tools/chain_check.py"""
import copy
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import oracle
from test_gpu_forward import _random_batch, seed0_model, seed0_state_dict, Bag

for sizes, fc in [([29] * 64, False), ([5, 6, 15, 16, 2, 31, 33, 64], False), ([65, 100, 40], True), ([70, 96, 12], False)]:
    max_atoms = 128 if max(sizes) > 100 or fc else 100
    m = copy.deepcopy(seed0_model(max_atoms=max_atoms)).cuda().eval()
    sd = seed0_state_dict(max_atoms=max_atoms)
    c = _random_batch(sizes, 99, fc)
    B = len(sizes)
    t = torch.arange(B, dtype=torch.float32) * 37.0 % 1000
    x, pos, ei, b = (torch.from_numpy(c[k]) for k in ("x", "pos", "edge_index", "batch"))
    with torch.no_grad():
        ea, ep, er = oracle.score_network_forward(sd, x, pos, ei, b, t, None, max_atoms=max_atoms, pad_to_max_atoms=False)
        ed = oracle.score_network_forward(sd.__class__({k: v.double() for k, v in sd.items()}), x.double(), pos.double(), ei, b,
                                          t.double(), None, max_atoms=max_atoms, pad_to_max_atoms=False) if os.environ.get("F64") else None
        a, p, r, _ = m(Bag(x=x.cuda(), pos=pos.cuda(), edge_index=ei.cuda(), batch=b.cuda()), t.cuda())
    line = f"sizes {str(sizes)[:30]:30s} fc={fc}: "
    for name, got, ref in (("atom", a, ea), ("pos", p, ep), ("prop", r, er)):
        d = (got.cpu() - ref).abs()
        line += f"{name} max abs {d.max():.2e} (|ref| max {ref.abs().max():.2f})  "
    print(line)
    if ed is not None:
        for name, got, ref32, ref in (("atom", a, ea, ed[0]), ("pos", p, ep, ed[1])):
            print(f"    vs float64 oracle: {name} hip {(got.cpu().double() - ref).abs().max():.2e}  float32 oracle {(ref32.double() - ref).abs().max():.2e}")
