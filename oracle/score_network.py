"""Oracle (test infrastructure): fp32 CPU restatement of the score-network forward.

Works on a plain ``state_dict`` (name -> fp32 tensor, the reference's key names) so that it
carries no module state.  Differentiable through torch autograd when the tensors in
``sd`` require grad, which is how the gradient parity tests get their expected values.
"""
import math

import torch
import torch.nn.functional as F


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def _ln(sd, name, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def fourier_time_embedding(t, dim):
    """model1.py:686-697: [sin(t f_k), cos(t f_k)], f_k = exp(-k ln(1e4)/(dim/2-1))."""
    half = dim // 2
    f = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1)))
    e = t.float().unsqueeze(-1) * f.unsqueeze(0)
    e = torch.cat([e.sin(), e.cos()], dim=-1)
    if e.shape[-1] < dim:
        e = F.pad(e, (0, dim - e.shape[-1]))
    return e


def pilot_guidance(sd, properties, null_conditioning, prefix="property_guidance"):
    """model1.py:335-359.  properties is [B, P] (P >= 8 in the parity domain)."""
    if null_conditioning or properties is None:
        b = 1 if properties is None else properties.shape[0]
        return sd[prefix + ".null_embedding"].unsqueeze(0).repeat(b, 1)
    n_obj = 8
    embs = []
    for i in range(min(n_obj, properties.shape[-1])):
        v = properties[:, i:i + 1]
        e = _lin(sd, f"{prefix}.property_encoders.{i}.2",
                 F.gelu(_lin(sd, f"{prefix}.property_encoders.{i}.0", v)))
        embs.append(e)
    while len(embs) < n_obj:
        embs.append(torch.zeros_like(embs[0]))
    stacked = torch.stack(embs, dim=1)                      # [B, 8, H]
    flat = stacked.reshape(stacked.shape[0], -1)
    w = F.softplus(_lin(sd, f"{prefix}.pareto_predictor.2",
                        F.gelu(_lin(sd, f"{prefix}.pareto_predictor.0", flat))))
    w = w / w.sum(dim=-1, keepdim=True)
    return (stacked * w.unsqueeze(-1)).sum(dim=1)


def e3_layer(sd, prefix, h, pos, edge_index):
    """model1.py:86-114 (+ :120-121 for the aggregation).

    Messages are summed at edge_index[1]; coordinates get -rel*w at edge_index[0] and
    +rel*w at edge_index[1].  The 1e-8 is added to each component before the norm.
    """
    row, col = edge_index[0], edge_index[1]
    rel = pos[col] - pos[row]
    dist = torch.norm(rel + 1e-8, dim=-1, keepdim=True)
    rbf = _lin(sd, prefix + ".radial_basis.2", F.silu(_lin(sd, prefix + ".radial_basis.0", dist)))
    ef = torch.cat([h[row], h[col], rbf], dim=-1)
    m = _lin(sd, prefix + ".edge_network.2", F.silu(_lin(sd, prefix + ".edge_network.0", ef)))
    m = _ln(sd, prefix + ".edge_network.3", m)
    agg = torch.zeros(h.shape[0], m.shape[-1], dtype=m.dtype).index_add(0, col, m)
    hn = _lin(sd, prefix + ".node_network.2",
              F.silu(_lin(sd, prefix + ".node_network.0", torch.cat([h, agg], dim=-1))))
    hn = _ln(sd, prefix + ".node_network.3", hn)
    cw = torch.tanh(_lin(sd, prefix + ".coord_network.2",
                         F.silu(_lin(sd, prefix + ".coord_network.0", m))))
    wrel = rel * cw
    dpos = torch.zeros_like(pos).index_add(0, row, -wrel).index_add(0, col, wrel)
    return hn, dpos, m


def dit_layer(sd, prefix, x, t_emb, key_pad, num_heads, drop=None):
    """model1.py:279-302 on a dense [B, M, H] batch; key_pad [B, M] is True at padding.

    nn.MultiheadAttention semantics (batch_first): packed in_proj, per-head scaled dot product
    with padded keys at -inf, out_proj.  drop (train mode, model1.py:263-274): dict of mask SCALES
    (0 or 1 / (1 - p), what torch.nn.functional.dropout multiplies by) - 'att' [B, heads, M, M] on the
    softmax probabilities, 'ffn_hidden' [B, M, 4H] after the GELU, 'ffn_out' [B, M, H] on the FFN
    output; a missing key means no dropout at that site.
    """
    drop = drop or {}
    B, M, H = x.shape
    dh = H // num_heads
    params = _lin(sd, prefix + ".adaLN_modulation.1", F.silu(t_emb))
    sh_a, sc_a, g_a, sh_m, sc_m, g_m = params.chunk(6, dim=-1)

    def ln(v):
        return F.layer_norm(v, (H,), None, None, 1e-6)

    xn = ln(x) * (1.0 + sc_a.unsqueeze(1)) + sh_a.unsqueeze(1)
    qkv = F.linear(xn, sd[prefix + ".attention.in_proj_weight"], sd[prefix + ".attention.in_proj_bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    q = q.view(B, M, num_heads, dh).transpose(1, 2)
    k = k.view(B, M, num_heads, dh).transpose(1, 2)
    v = v.view(B, M, num_heads, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    if key_pad is not None:
        s = s.masked_fill(key_pad[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    if "att" in drop:
        p = p * drop["att"]
    o = (p @ v).transpose(1, 2).reshape(B, M, H)
    o = _lin(sd, prefix + ".attention.out_proj", o)
    x = x + g_a.unsqueeze(1) * o
    xn = ln(x) * (1.0 + sc_m.unsqueeze(1)) + sh_m.unsqueeze(1)
    hid = F.gelu(_lin(sd, prefix + ".ff_network.0", xn))
    if "ffn_hidden" in drop:
        hid = hid * drop["ffn_hidden"]
    ff = _lin(sd, prefix + ".ff_network.3", hid)
    if "ffn_out" in drop:
        ff = ff * drop["ffn_out"]
    return x + g_m.unsqueeze(1) * ff


def _head(sd, prefix, h):
    """model1.py:636-655: Linear, GELU, LayerNorm, Linear."""
    y = _ln(sd, prefix + ".2", F.gelu(_lin(sd, prefix + ".0", h)))
    return _lin(sd, prefix + ".3", y)


def score_network_forward(sd, x, pos, edge_index, batch, t, properties=None,
                          classifier_free_guidance=False, max_atoms=100, num_heads=8,
                          pad_to_max_atoms=True, return_intermediates=False, dropout=None):
    """model1.py:699-755 without the consistency module; eval mode unless ``dropout`` is given.

    dropout: one dict per attention layer with the mask scales of its three dropout sites in
    RAGGED form - 'att' [N, heads, K] (query atom, head, key index within the atom's molecule; K >=
    the largest molecule), 'ffn_hidden' [N, 4H], 'ffn_out' [N, H] - as ligdiff_dropout_mask exports
    them; they are scattered into the dense layout here.

    pad_to_max_atoms=True reproduces the reference's to_dense_batch(max_num_nodes=
    max_atoms) layout (model1.py:728); False pads only to the longest molecule, which is
    result-equivalent for valid atoms because padded keys are masked (model1.py:294,
    :736) and padded query rows are dropped (model1.py:738).
    Returns atom_pred [N,A], pos_pred [N,3], prop_pred [B,P].
    """
    H = sd["time_mlp.0.weight"].shape[1]
    B = int(batch.max().item()) + 1
    inter = {}
    t_emb = fourier_time_embedding(t, H)
    t_emb = _lin(sd, "time_mlp.2", F.gelu(_lin(sd, "time_mlp.0", t_emb)))
    h = F.gelu(_ln(sd, "atom_embedding.1", _lin(sd, "atom_embedding.0", x)))
    h = h + F.gelu(_ln(sd, "pos_embedding.1", _lin(sd, "pos_embedding.0", pos))) + t_emb[batch]
    if properties is not None:
        h = h + pilot_guidance(sd, properties, classifier_free_guidance)[batch]
    inter["h_embed"] = h
    n_e3 = len({k.split(".")[1] for k in sd if k.startswith("equivariant_layers.")})
    n_dit = len({k.split(".")[1] for k in sd if k.startswith("dit_layers.")})
    pos_updates = torch.zeros_like(pos)
    for l in range(n_e3):
        hn, dpos, m = e3_layer(sd, f"equivariant_layers.{l}", h, pos + pos_updates, edge_index)
        h = h + hn
        pos_updates = pos_updates + dpos * 0.1
        inter[f"edge_msg_{l}"] = m
        inter[f"h_e3_{l}"] = h
    inter["pos_updates"] = pos_updates
    counts = torch.bincount(batch, minlength=B)
    if int(counts.max()) > max_atoms:
        raise ValueError("molecule larger than max_atoms")
    M = max_atoms if pad_to_max_atoms else int(counts.max())
    starts = torch.cumsum(counts, 0) - counts
    slot = torch.arange(batch.numel()) - starts[batch]
    dense = h.new_zeros(B, M, H)
    mask = torch.zeros(B, M, dtype=torch.bool)
    dense[batch, slot] = h
    mask[batch, slot] = True
    if t_emb.shape[0] >= B:
        t_b = t_emb[:B]
    else:
        t_b = t_emb[0].repeat(B, 1)                            # model1.py:729-733
    for l in range(n_dit):
        drop = None
        if dropout is not None:
            drop = {}
            for key, val in dropout[l].items():
                if key == "att":
                    d = val.new_ones(B, M, val.shape[1], M)
                    d[batch, slot] = val[:, :, :M] if val.shape[2] >= M else F.pad(val, (0, M - val.shape[2]), value=1.0)
                    drop[key] = d.permute(0, 2, 1, 3)                  # [B, heads, M(query), M(key)]
                else:
                    d = val.new_ones(B, M, val.shape[1])
                    d[batch, slot] = val
                    drop[key] = d
        dense = dit_layer(sd, f"dit_layers.{l}", dense, t_b, ~mask, num_heads, drop)
    h = dense[mask]
    inter["h_dit"] = h
    atom_pred = _head(sd, "atom_output", h)
    pos_pred = _head(sd, "pos_output", h) + pos_updates
    pooled = torch.zeros(B, H, dtype=h.dtype).index_add(0, batch, h) / counts.clamp(min=1).unsqueeze(-1).to(h.dtype)
    prop_pred = _head(sd, "property_output", pooled)
    if return_intermediates:
        return atom_pred, pos_pred, prop_pred, inter
    return atom_pred, pos_pred, prop_pred
