"""Writes tests/golden/layers.npz by RUNNING THE REFERENCE in this container (python tests/golden/make_golden_layers.py):
models with other layer counts than the default.  model1.py:627-634 builds num_layers // 2 E(3) layers and as many
attention layers; main.py:242 asks for num_layers = 8, the smoke block of generation.py:992 for 2.

Per layer count (8 -> four + four layers, 2 -> one + one): per-tensor checksums of the seed-0 weights, an eval forward
on a ragged template batch with properties, and the trainer's loss step (dropout probabilities 0, recorded draws) with
the loss terms, the gradient norm and per-parameter gradient sums (small gradients in full), as in make_golden.py."""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_loader  # noqa: E402
from make_golden import build_model, checksums, run_checked, template_batch  # noqa: E402

torch.set_num_threads(8)


def main():
    m1, tr, gen = _ref_loader.load_reference()
    Data = sys.modules["torch_geometric.data"].Data
    G = gen.ResearchValidatedGenerator
    out = {}
    for nl in (8, 2):
        tag = f"L{nl}"
        g = torch.Generator().manual_seed(9000 + nl)
        model = build_model(m1, num_layers=nl).eval()
        assert len(model.equivariant_layers) == nl // 2 and len(model.dit_layers) == nl // 2
        for k, v in checksums(model.state_dict()).items():
            if k.startswith(("equivariant_layers.", "dit_layers.", "atom_output.")):
                out[f"{tag}_ck_{k}"] = v
        sizes = [7, 19, 29, 11, 40]
        x, pos, ei, batch = template_batch(G, sizes, g)
        B = len(sizes)
        props = torch.rand(B, 15, generator=g)
        t = torch.tensor([900, 31, 512, 0, 250])
        with torch.no_grad():
            a, pp, pr, cons = run_checked(lambda: model(Data(x=x, pos=pos, edge_index=ei, batch=batch), t, props, False))
        assert cons is None and float(a.abs().sum()) > 0
        out.update({f"{tag}_sizes": np.array(sizes), f"{tag}_x": x.numpy(), f"{tag}_pos": pos.numpy(),
                    f"{tag}_edge_index": ei.numpy(), f"{tag}_batch": batch.numpy(), f"{tag}_props": props.numpy(),
                    f"{tag}_t": t.numpy(), f"{tag}_atom": a.numpy(), f"{tag}_pos_pred": pp.numpy(), f"{tag}_prop": pr.numpy()})
        # trainer loss step + gradients (dropout p = 0)
        tmodel = build_model(m1, num_layers=nl)
        for lyr in tmodel.dit_layers:
            lyr.attention.dropout = 0.0
            lyr.ff_network[2].p = 0.0
            lyr.ff_network[4].p = 0.0
        with contextlib.redirect_stdout(io.StringIO()):
            trainer = tr.ResearchAccurateTrainer(tmodel, [], [], [], debug_mode=True)
        tmodel.train()
        sizes_t = [6, 9, 5, 7]
        xt, post, eit, bt = template_batch(G, sizes_t, g)
        xt = torch.nn.functional.one_hot(torch.randint(0, 10, (xt.shape[0],), generator=g), 119).float()
        Bt = len(sizes_t)
        props_t = torch.rand(Bt, 15, generator=g)
        batch_obj = Data(x=xt, pos=post, edge_index=eit, edge_attr=None, batch=bt)
        batch_obj.properties = props_t.clone()
        seed = 4100 + nl
        torch.manual_seed(seed)
        u = torch.cat([torch.rand(1) for _ in range(Bt)])
        noise = torch.randn_like(post)
        null_rand = torch.rand(Bt)
        torch.manual_seed(seed)
        tmodel.zero_grad()
        losses = run_checked(lambda: trainer._compute_research_exact_loss(batch_obj, training=True))
        losses["total_loss"].backward()
        for k, v in losses.items():
            if torch.is_tensor(v):
                out[f"{tag}_loss_{k}"] = np.float64(v.item())
        gsq = 0.0
        for name, p in tmodel.named_parameters():
            if p.grad is not None:
                gd = p.grad.double()
                gsq += float((gd ** 2).sum())
                out[f"{tag}_gradsum_{name}"] = np.array([float(gd.sum()), float(gd.abs().sum()), float((gd ** 2).sum())])
                if p.numel() <= 4096:
                    out[f"{tag}_grad_{name}"] = p.grad.numpy().copy()
        out[f"{tag}_gradnorm"] = np.float64(gsq ** 0.5)
        out.update({f"{tag}_tsizes": np.array(sizes_t), f"{tag}_tx": xt.numpy(), f"{tag}_tpos": post.numpy(),
                    f"{tag}_tedge_index": eit.numpy(), f"{tag}_tbatch": bt.numpy(), f"{tag}_tprops": props_t.numpy(),
                    f"{tag}_u": u.numpy(), f"{tag}_noise": noise.numpy(), f"{tag}_null_rand": null_rand.numpy()})
    np.savez_compressed(os.path.join(HERE, "layers.npz"), **out)
    print("wrote layers.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
