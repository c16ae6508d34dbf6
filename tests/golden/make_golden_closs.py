"""Writes tests/golden/compute_loss.npz by RUNNING THE REFERENCE's
ResearchAccurateDiffusionModel.compute_loss (model1.py:767-888) in this container (through
tests/golden/_ref_loader.py; needs /root/reference).  Run from the repo root:
    python tests/golden/make_golden_closs.py
Eval mode (no dropout, no consistency outputs), seed-0 weights, a ragged template batch, without and
with properties.  The random draws the reference makes (sigma uniforms, coordinate noise, the
null-conditioning uniforms - in that order, model1.py:786, :789, :833) are replayed from the same seed
and stored, together with every loss value and per-parameter gradient checksums of losses['total']."""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_loader  # noqa: E402
from make_golden import build_model, template_batch  # noqa: E402


def main():
    m1, tr, gen = _ref_loader.load_reference()
    Data = sys.modules["torch_geometric.data"].Data
    G = gen.ResearchValidatedGenerator
    model = build_model(m1).eval()
    g = torch.Generator().manual_seed(2468)
    sizes = [6, 11, 5, 17, 8]
    x, pos, ei, batch = template_batch(G, sizes, g)
    x = torch.nn.functional.one_hot(torch.randint(0, 10, (x.shape[0],), generator=g), 119).float()
    B = len(sizes)
    props = torch.rand(B, 15, generator=g)
    out = dict(sizes=np.array(sizes), x=x.numpy(), pos=pos.numpy(), edge_index=ei.numpy(), batch=batch.numpy(),
               props=props.numpy())
    for tag, p, seed, weights in (("noprop", None, 91, None), ("prop", props, 92, None),
                                  ("prop_w", props, 93, {"coordinate": 2.0, "atom_type": 0.5, "property": 0.7,
                                                         "valency": 0.4, "adjacency": 0.6})):
        torch.manual_seed(seed)
        u = torch.rand(B)
        noise = torch.randn_like(pos)
        null_rand = torch.rand(B) if p is not None else torch.zeros(B)
        torch.manual_seed(seed)
        model.zero_grad()
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            losses, metrics = model.compute_loss(Data(x=x, pos=pos, edge_index=ei, edge_attr=None, batch=batch),
                                                 None, p, weights)
        assert "failed" not in buf.getvalue().lower(), buf.getvalue()
        losses["total"].backward()
        for k, v in losses.items():
            out[f"{tag}_loss_{k}"] = np.float64(v.item())
        for k, v in metrics.items():
            out[f"{tag}_metric_{k}"] = np.float64(v)
        for name, prm in model.named_parameters():
            if prm.grad is not None:
                gd = prm.grad.double()
                out[f"{tag}_gradsum_{name}"] = np.array([float(gd.sum()), float(gd.abs().sum()), float((gd ** 2).sum())])
                if prm.numel() <= 4096:
                    out[f"{tag}_grad_{name}"] = prm.grad.numpy().copy()
        out[f"{tag}_u"], out[f"{tag}_noise"], out[f"{tag}_null_rand"] = u.numpy(), noise.numpy(), null_rand.numpy()
        if weights is not None:
            out[f"{tag}_weights"] = np.array([weights[k] for k in ("coordinate", "atom_type", "property", "valency", "adjacency")])
        print(tag, {k: float(v) for k, v in losses.items()})
    np.savez_compressed(os.path.join(HERE, "compute_loss.npz"), **out)
    print("written compute_loss.npz")


if __name__ == "__main__":
    main()
