"""Writes tests/golden/quirk.npz by RUNNING THE REFERENCE in this container (python tests/golden/make_golden_quirk.py):
the two shapes the reference's own data pipeline produces that the other fixtures avoid.

1. ``properties`` as data_utils.py:904-914 stores them - one 1-D vector of 14 values per molecule - which PyG's
   collate concatenates to [14 B].  train.py:327-337 turns that into [B, 14 B] (unsqueeze + repeat): every molecule is
   conditioned on molecule 0's first eight values, and the PILOT loss family is skipped because
   F.mse_loss(prop_pred [B,15], properties [B,14 B]) raises inside the try block of train.py:388-397.
2. A one-atom molecule with the [[0, 0]] self-loop edge of data_utils.py:893-895, and one with no edge at all,
   next to ordinary molecules: eval forward outputs and the same train-mode loss step.
3. Dict-valued ``target_properties`` (the scenarios of generation.py:689-757): the model call fails on them and
   returns zeros (model1.py:337, :757-764); DDPM trajectories of generation.py:393-484 with such properties, with
   and without guidance.
Dropout probabilities are 0 as in make_golden.py; the random draws of the loss step are recorded."""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_loader  # noqa: E402
from make_golden import build_model  # noqa: E402

torch.set_num_threads(8)


def batch_of(G, sizes, g, lone_edge):
    """Template molecules; a one-atom molecule gets the self-loop [[0, 0]] (lone_edge) or no edge."""
    xs, ps, es, bs, off = [], [], [], [], 0
    for m, n in enumerate(sizes):
        if n == 1:
            ei = torch.zeros(2, 1, dtype=torch.long) if lone_edge else torch.zeros(2, 0, dtype=torch.long)
        else:
            ei = torch.tensor(G.create_research_connectivity_fixed(None, n)[0], dtype=torch.long).t()
        xs.append(torch.nn.functional.one_hot(torch.randint(0, 10, (n,), generator=g), 119).float())
        ps.append(torch.randn(n, 3, generator=g) * 1.5)
        es.append(ei + off)
        bs.append(torch.full((n,), m, dtype=torch.long))
        off += n
    return torch.cat(xs), torch.cat(ps), torch.cat(es, 1), torch.cat(bs)


def main():
    m1, tr, gen = _ref_loader.load_reference()
    Data = sys.modules["torch_geometric.data"].Data
    G = gen.ResearchValidatedGenerator
    g = torch.Generator().manual_seed(4242)
    out = {}
    sizes = [6, 1, 9, 1, 5]
    out["sizes"] = np.array(sizes)
    B = len(sizes)
    # ---- eval forward with one-atom molecules (self-loop on the first, no edge on the second)
    x, pos, _, batch = batch_of(G, sizes, g, True)
    parts, off = [], 0
    for m, n in enumerate(sizes):
        if n == 1:
            ei = torch.zeros(2, 1, dtype=torch.long) if m == 1 else torch.zeros(2, 0, dtype=torch.long)
        else:
            ei = torch.tensor(G.create_research_connectivity_fixed(None, n)[0], dtype=torch.long).t()
        parts.append(ei + off)
        off += n
    ei = torch.cat(parts, 1)
    model = build_model(m1).eval()
    t = torch.tensor([17.0, 250.0, 999.0, 3.0, 640.0])
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), torch.no_grad():
        a, p, r, _ = model(Data(x=x, pos=pos, edge_index=ei, edge_attr=None, batch=batch), t)
    assert "error" not in buf.getvalue() and "failed" not in buf.getvalue(), buf.getvalue()
    out.update(fwd_x=x.numpy(), fwd_pos=pos.numpy(), fwd_edge_index=ei.numpy(), fwd_batch=batch.numpy(), fwd_t=t.numpy(),
               fwd_atom=a.numpy(), fwd_pos_pred=p.numpy(), fwd_prop=r.numpy())

    # ---- loss step on a batch shaped like the reference's own loader output
    tmodel = build_model(m1)
    for lyr in tmodel.dit_layers:
        lyr.attention.dropout = 0.0
        lyr.ff_network[2].p = 0.0
        lyr.ff_network[4].p = 0.0
    with contextlib.redirect_stdout(io.StringIO()):
        trainer = tr.ResearchAccurateTrainer(tmodel, [], [], [], debug_mode=True)
    tmodel.train()
    xt, post, eit, bt = batch_of(G, sizes, g, True)                     # both one-atom molecules carry the self-loop
    props_flat = torch.rand(14 * B, generator=g)                       # PyG collate of B vectors of 14 values
    batch_obj = Data(x=xt, pos=post, edge_index=eit, edge_attr=None, batch=bt)
    batch_obj.properties = props_flat.clone()
    seed = 909
    torch.manual_seed(seed)                                             # replay the reference's draw order
    u = torch.cat([torch.rand(1) for _ in range(B)])
    noise = torch.randn_like(post)
    null_rand = torch.rand(B)
    torch.manual_seed(seed)
    tmodel.zero_grad()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        losses = trainer._compute_research_exact_loss(batch_obj, training=True)
    txt = buf.getvalue()
    # the one failure the reference is expected to swallow here is the PILOT shape mismatch
    lines = [ln for ln in txt.splitlines() if "failed" in ln or "error" in ln.lower()]
    assert len(lines) == 1 and lines[0].startswith("PILOT loss computation failed"), txt
    out["pilot_failure_message"] = np.array(lines[0])
    losses["total_loss"].backward()
    for k, v in losses.items():
        if torch.is_tensor(v):
            out[f"loss_{k}"] = np.float64(v.item())
    out["loss_keys"] = np.array(sorted(losses.keys()))
    gsq = 0.0
    for name, p_ in tmodel.named_parameters():
        if p_.grad is not None:
            gd = p_.grad.double()
            gsq += float((gd ** 2).sum())
            out[f"gradsum_{name}"] = np.array([float(gd.sum()), float(gd.abs().sum()), float((gd ** 2).sum())])
            if p_.numel() <= 4096:
                out[f"grad_{name}"] = p_.grad.numpy().copy()
    out["gradnorm"] = np.float64(gsq ** 0.5)
    out.update(x=xt.numpy(), pos=post.numpy(), edge_index=eit.numpy(), batch=bt.numpy(), props_flat=props_flat.numpy(),
               u=u.numpy(), noise=noise.numpy(), null_rand=null_rand.numpy())
    # ---- DDPM with dict-valued target properties (seeded draws in the reference's call order)
    for tag, scale in (("dict_guided", 2.0), ("dict_plain", 1.0)):
        T, n, seed = 20, 13, 2468
        smodel = build_model(m1, timesteps=T).eval()
        with contextlib.redirect_stdout(io.StringIO()):
            generator = G(smodel)
        final = {}

        def capture(x_, pos_, n_, _fin=final):
            _fin["x"], _fin["pos"] = x_.clone(), pos_.clone()
            return True

        generator._validate_final_molecule = capture
        tp = {"molecular_weight": 350.0, "logp": 2.5, "qed": 0.8}
        torch.manual_seed(seed)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf), torch.no_grad():
            res = generator.ddpm_sampling_fixed(n, tp, scale, 0.9)
        assert res is not None and "Forward pass failed" in buf.getvalue(), buf.getvalue()[:300]
        out[f"{tag}_meta"] = np.array([T, n, seed])
        out[f"{tag}_scale"] = np.float32(scale)
        out[f"{tag}_final_x"] = final["x"].numpy()
        out[f"{tag}_final_pos"] = final["pos"].numpy()
    np.savez_compressed(os.path.join(HERE, "quirk.npz"), **out)
    print("wrote quirk.npz:", {k: float(v) for k, v in out.items() if k.startswith("loss_") and k != "loss_keys"})
    print("parameters with a gradient:", sum(1 for k in out if k.startswith("gradsum_")), "message:", lines[0])


if __name__ == "__main__":
    main()
