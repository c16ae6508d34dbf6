"""Writes tests/golden/dropout.npz by RUNNING THE REFERENCE's trainer loss step in TRAIN mode with its
dropout ACTIVE (p = 0.1: nn.MultiheadAttention's attention-probability dropout and the two nn.Dropout of
ff_network, model1.py:263-274) in this container (through tests/golden/_ref_loader.py; needs
/root/reference).  Run from the repo root:  python tests/golden/make_golden_dropout.py

torch's dropout stream cannot be reproduced on another device, so the masks are made visible instead:
torch.nn.functional.dropout is replaced, for the duration of the run, by a function that draws the keep
mask from a seeded generator, records it and returns x * mask / (1 - p) - exactly what the original
computes for its own mask.  Every call site of the attention layers goes through that name
(multi_head_attention_forward and nn.Dropout.forward both call torch.nn.functional.dropout).  The fixture
holds the inputs, the recorded masks (bit-packed, in call order), the reference's loss values and
per-parameter gradient checksums.  The model uses max_atoms = 16 so that the padded masks stay small."""
import contextlib
import io
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_loader  # noqa: E402
from make_golden import build_model, template_batch  # noqa: E402


def main():
    m1, tr, gen = _ref_loader.load_reference()
    Data = sys.modules["torch_geometric.data"].Data
    G = gen.ResearchValidatedGenerator
    model = build_model(m1, max_atoms=16)
    with contextlib.redirect_stdout(io.StringIO()):
        trainer = tr.ResearchAccurateTrainer(model, [], [], [], debug_mode=True)
    model.train()
    g = torch.Generator().manual_seed(97531)
    sizes = [6, 13, 5, 16]
    x, pos, ei, batch = template_batch(G, sizes, g)
    x = torch.nn.functional.one_hot(torch.randint(0, 10, (x.shape[0],), generator=g), 119).float()
    B = len(sizes)
    props = torch.rand(B, 15, generator=g)
    out = dict(sizes=np.array(sizes), x=x.numpy(), pos=pos.numpy(), edge_index=ei.numpy(), batch=batch.numpy(),
               props=props.numpy(), max_atoms=np.int64(16))

    mask_gen = torch.Generator().manual_seed(424242)
    calls = []
    real_dropout = F.dropout

    def recording_dropout(input, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return input
        keep = torch.rand(input.shape, generator=mask_gen) >= p
        calls.append((tuple(input.shape), float(p), keep))
        return input * keep.to(input.dtype) / (1.0 - p)

    for tag, with_props, seed in (("noprop", False, 1357), ("prop", True, 2468)):
        batch_obj = Data(x=x, pos=pos, edge_index=ei, edge_attr=None, batch=batch)
        if with_props:
            batch_obj.properties = props.clone()
        torch.manual_seed(seed)                    # the trainer's own draws, replayed in its order
        u = torch.cat([torch.rand(1) for _ in range(B)])
        noise = torch.randn_like(pos)
        null_rand = torch.rand(B) if with_props else torch.zeros(B)
        torch.manual_seed(seed)
        model.zero_grad()
        calls.clear()
        F.dropout = recording_dropout
        try:
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                losses = trainer._compute_research_exact_loss(batch_obj, training=True)
        finally:
            F.dropout = real_dropout
        txt = buf.getvalue()
        assert "failed" not in txt.lower() and "error" not in txt.lower(), txt
        losses["total_loss"].backward()
        # call order per attention layer: attention probabilities [B*heads, M, M], FFN hidden [B, M, 1024],
        # FFN output [B, M, 256]; the consistency module's two dropouts follow (its outputs reach no loss)
        dit = [c for c in calls if c[0][-1] in (16, 1024, 256) and len(c[0]) == 3][:9]
        assert len(dit) == 9, [c[0] for c in calls]
        for l in range(3):
            att, hid, ffo = dit[3 * l:3 * l + 3]
            assert att[0] == (B * 8, 16, 16) and hid[0] == (B, 16, 1024) and ffo[0] == (B, 16, 256), (att[0], hid[0], ffo[0])
            assert att[1] == hid[1] == ffo[1] == 0.1
            out[f"{tag}_mask_att_{l}"] = np.packbits(att[2].numpy().reshape(-1))
            out[f"{tag}_mask_hid_{l}"] = np.packbits(hid[2].numpy().reshape(-1))
            out[f"{tag}_mask_out_{l}"] = np.packbits(ffo[2].numpy().reshape(-1))
        for k, v in losses.items():
            if torch.is_tensor(v):
                out[f"{tag}_loss_{k}"] = np.float64(v.item())
        gsq = 0.0
        for name, prm in model.named_parameters():
            if prm.grad is not None:
                gd = prm.grad.double()
                gsq += float((gd ** 2).sum())
                out[f"{tag}_gradsum_{name}"] = np.array([float(gd.sum()), float(gd.abs().sum()), float((gd ** 2).sum())])
                if prm.numel() <= 1024:
                    out[f"{tag}_grad_{name}"] = prm.grad.numpy().copy()
        out[f"{tag}_gradnorm"] = np.float64(gsq ** 0.5)
        out[f"{tag}_u"], out[f"{tag}_noise"], out[f"{tag}_null_rand"] = u.numpy(), noise.numpy(), null_rand.numpy()
        print(tag, "total", float(losses["total_loss"]), "grad norm", gsq ** 0.5, "dropout calls", len(calls))
    np.savez_compressed(os.path.join(HERE, "dropout.npz"), **out)
    print("written dropout.npz", os.path.getsize(os.path.join(HERE, "dropout.npz")), "bytes")


if __name__ == "__main__":
    main()
