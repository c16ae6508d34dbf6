"""Oracle (test infrastructure): the MolDiff consistency module, model1.py:453-561.

Same arithmetic as the reference with the per-pair Python loop of model1.py:513-540 written as
one batched MLP over all ordered pairs (i, j != i) of every molecule, in the loop's order.
Dropout (model1.py:390, :401) is the identity unless mask scales are passed in.
"""
import torch
import torch.nn.functional as F


def _mlp(sd, pre, x, last, mask=None):
    """Linear, GELU, LayerNorm, Linear, GELU, [Dropout], Linear (model1.py:384-414); mask [rows,128]:
    the Dropout's scales (0 or 1 / (1 - p))."""
    x = F.gelu(F.linear(x, sd[pre + ".0.weight"], sd[pre + ".0.bias"]))
    x = F.layer_norm(x, (256,), sd[pre + ".2.weight"], sd[pre + ".2.bias"])
    x = F.gelu(F.linear(x, sd[pre + ".3.weight"], sd[pre + ".3.bias"]))
    if mask is not None:
        x = x * mask
    return F.linear(x, sd[pre + f".{last}.weight"], sd[pre + f".{last}.bias"])


def consistency_forward(sd, h, pos, batch, dropout=None):
    """dropout: {'atom': [N,128], 'bond': [P,128]} mask scales of the two Dropout layers (train mode).
    h [N,256]: hidden state after the attention layers; pos [N,3]: data.pos + pos_updates;
    batch [N] sorted.  Returns (atom_logits [N,119], bond_logits [P,5] or None,
    valency_constrained [N,N], adjacency [B,n_max,n_max,5]) as model1.py:478."""
    P = "moldiff_consistency."
    dropout = dropout or {}
    atom_logits = _mlp(sd, P + "atom_predictor", h, 6, dropout.get("atom"))   # model1.py:458
    atom_probs = F.softmax(atom_logits, dim=-1)
    valency_pred = F.softplus(_mlp(sd, P + "valency_predictor", torch.cat([h, atom_probs], -1), 5))
    predicted = torch.argmax(atom_probs, dim=-1)
    max_allowed = sd[P + "max_valences"][predicted.clamp(0, 118)]
    valency_constrained = torch.min(valency_pred.squeeze(), max_allowed.unsqueeze(-1))   # [N] vs [N,1] -> [N,N]
    B = int(batch.max().item()) + 1
    counts = torch.bincount(batch, minlength=B)
    n_max = int(counts.max().item())
    radii = sd[P + "covalent_radii"]
    adj = torch.zeros(B, n_max, n_max, 5)
    logits_all = []
    start = 0
    pair0 = 0
    for m in range(B):
        n = int(counts[m])
        idx = torch.arange(n)
        adj[m, idx, idx, 0] = 1.0                                             # model1.py:515-518
        if n > 1:
            ii, jj = torch.meshgrid(idx, idx, indexing="ij")
            keep = ii != jj
            ii, jj = ii[keep], jj[keep]                                       # row-major, the loop's order
            hi, hj = h[start + ii], h[start + jj]
            d = (pos[start + ii] - pos[start + jj]).norm(dim=-1)
            L = radii[predicted[start + ii].clamp(max=118)] + radii[predicted[start + jj].clamp(max=118)]
            feats = torch.stack([d, L, d / L, (d < 3.0).float()], dim=-1)     # model1.py:528-533
            bm = dropout["bond"][pair0:pair0 + ii.numel()] if "bond" in dropout else None
            pair0 += ii.numel()
            lg = _mlp(sd, P + "bond_predictor", torch.cat([hi, hj, feats], -1), 6, bm)
            adj[m, ii, jj] = F.softmax(lg, dim=-1)
            logits_all.append(lg)
        start += n
    bond_logits = torch.cat(logits_all) if logits_all else None
    return atom_logits, bond_logits, valency_constrained, adj
