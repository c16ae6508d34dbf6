"""Fixtures for the two alternate samplers (model1.py:890-966 ``sample`` and train.py:786-863
``generate_research_validated_molecules``), written by RUNNING THE REFERENCE in this container.

Run from the repo root:  python tests/golden/make_golden_alt.py
Both reference functions draw from the global torch generator; the fixture stores the seed and
the draws replayed in the reference's order (per molecule: coordinates, features, edge
attributes, then one coordinate-noise tensor per noisy step), plus the coordinates (and the last
atom prediction) the reference hands to its RDKit post-processing, which is stubbed out.
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_loader  # noqa: E402
from make_golden import build_model, run_checked  # noqa: E402

torch.set_num_threads(8)


def main():
    m1, tr, gen = _ref_loader.load_reference()
    out = {}
    g = torch.Generator().manual_seed(99)
    tp = torch.rand(1, 15, generator=g)
    out["props"] = tp.numpy()

    # ---- model1.sample ---------------------------------------------------------------
    model = build_model(m1).eval()
    for tag, props, scale in (("ddim_plain", None, 1.0), ("ddim_cfg", tp, 2.0)):
        S, n, steps, seed = 2, 14, 12, 2468
        got = []
        model._postprocess_molecule = lambda data, atom_pred, _g=got: (_g.append(
            (data.pos.clone(), atom_pred.clone())) or {"smiles": "X"})
        torch.manual_seed(seed)
        with torch.no_grad():
            run_checked(lambda: model.sample(num_samples=S, target_properties=props, guidance_scale=scale,
                                             num_steps=steps, max_atoms=n))
        assert len(got) == S
        # replay the draws
        torch.manual_seed(seed)
        ts = torch.linspace(model.timesteps - 1, 0, steps, dtype=torch.long)
        x0, p0, nz = [], [], []
        for _ in range(S):
            p0.append(torch.randn(n, 3))
            x0.append(torch.randn(n, 119))
            torch.randn(2 * (n - 1), 5)
            zs = []
            for i in range(steps):
                if i < steps - 1 and ts[i + 1] > 0:
                    zs.append(torch.randn(n, 3))
                else:
                    zs.append(torch.zeros(n, 3))
            nz.append(torch.stack(zs))
        out[f"{tag}_meta"] = np.array([S, n, steps, seed])
        out[f"{tag}_scale"] = np.float32(scale)
        out[f"{tag}_x0"] = torch.stack(x0).numpy()
        out[f"{tag}_pos0"] = torch.stack(p0).numpy()
        out[f"{tag}_noise"] = torch.stack(nz).numpy()
        out[f"{tag}_pos"] = torch.stack([a for a, _ in got]).numpy()
        out[f"{tag}_atom_pred"] = torch.stack([b for _, b in got]).numpy()

    # ---- trainer Euler sampler ---------------------------------------------------------
    tmodel = build_model(m1)
    with contextlib.redirect_stdout(io.StringIO()):
        trainer = tr.ResearchAccurateTrainer(tmodel, [], [], [], debug_mode=True)
    for tag, props, scale in (("euler_plain", None, 7.5), ("euler_cfg", tp, 3.0)):
        S, n, steps, seed, temp = 2, 25, 10, 1357, 0.8
        got = []
        trainer._postprocess_to_valid_molecule = lambda data, target_properties, _g=got: (_g.append(
            data.pos.clone()) or {"smiles": "X"})
        torch.manual_seed(seed)
        buf = io.StringIO()
        with torch.no_grad(), contextlib.redirect_stdout(buf):
            res = trainer.generate_research_validated_molecules(num_samples=S, target_properties=props,
                                                                guidance_scale=scale, num_steps=steps,
                                                                temperature=temp)
        assert len(res) == S and len(got) == S, buf.getvalue()
        assert "Generation failed" not in buf.getvalue(), buf.getvalue()
        torch.manual_seed(seed)
        E = trainer._create_initial_graph(n).shape[1]
        x0, p0, nz = [], [], []
        for _ in range(S):
            p0.append(torch.randn(n, 3) * 2.0)
            x0.append(torch.randn(n, 119))
            torch.randn(E, 5)
            zs = []
            for step in range(steps):
                zs.append(torch.randn(n, 3) if step < steps - 1 else torch.zeros(n, 3))
            nz.append(torch.stack(zs))
        out[f"{tag}_meta"] = np.array([S, n, steps, seed])
        out[f"{tag}_scale"] = np.float32(scale)
        out[f"{tag}_temperature"] = np.float32(temp)
        out[f"{tag}_edges"] = trainer._create_initial_graph(n).numpy()
        out[f"{tag}_x0"] = torch.stack(x0).numpy()
        out[f"{tag}_pos0"] = torch.stack(p0).numpy()
        out[f"{tag}_noise"] = torch.stack(nz).numpy()
        out[f"{tag}_pos"] = torch.stack(got).numpy()
    np.savez_compressed(os.path.join(HERE, "alt_samplers.npz"), **out)
    print("written", os.path.join(HERE, "alt_samplers.npz"))


if __name__ == "__main__" and "--consistency" not in sys.argv:
    main()


def consistency_fixture():
    """Train-mode forward of the reference with every dropout at p = 0: the 4th return value
    (MolDiff consistency outputs, model1.py:750-753).  The two chemistry buffers are stored as
    inputs: without RDKit they hold the loader's placeholder table."""
    from make_golden import template_batch
    m1, tr, gen = _ref_loader.load_reference()
    Data = sys.modules["torch_geometric.data"].Data
    model = build_model(m1)
    for lyr in model.dit_layers:
        lyr.attention.dropout = 0.0
        lyr.ff_network[2].p = 0.0
        lyr.ff_network[4].p = 0.0
    model.moldiff_consistency.atom_predictor[5].p = 0.0
    model.moldiff_consistency.bond_predictor[5].p = 0.0
    model.train()
    g = torch.Generator().manual_seed(31)
    sizes = [5, 9, 2, 7]
    x, pos, ei, b = template_batch(gen.ResearchValidatedGenerator, sizes, g)
    t = torch.tensor([10.0, 500.0, 900.0, 42.0])
    with torch.no_grad():
        out = run_checked(lambda: model(Data(x=x, pos=pos, edge_index=ei, edge_attr=None, batch=b), t))
    atom_pred, pos_pred, prop_pred, cons = out
    assert cons is not None and cons[1] is not None
    np.savez_compressed(
        os.path.join(HERE, "consistency.npz"), sizes=np.array(sizes), x=x.numpy(), pos=pos.numpy(),
        edge_index=ei.numpy(), batch=b.numpy(), t=t.numpy(), atom_pred=atom_pred.numpy(), pos_pred=pos_pred.numpy(),
        max_valences=model.moldiff_consistency.max_valences.numpy(),
        covalent_radii=model.moldiff_consistency.covalent_radii.numpy(),
        atom_logits=cons[0].numpy(), bond_logits=cons[1].numpy(), valency=cons[2].numpy(), adjacency=cons[3].numpy())
    print("written consistency.npz", [tuple(c.shape) for c in cons])


if __name__ == "__main__" and "--consistency" in sys.argv:
    consistency_fixture()
