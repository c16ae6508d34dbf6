"""Oracle (test infrastructure): the EDM noise / loss step of the trainer.

Restates train.py:267-452 with its loss families (:454-473 EDM, :475-498 PILOT, :603-623
DiT, :625-663 E(3)).  The random draws of the reference (one uniform per molecule for
sigma at :289-293, Gaussian coordinate noise at :305, one uniform per molecule for the
null-conditioning mask at :341) are inputs, so the HIP path can be fed the same numbers.
The MolDiff family (:500-601) is gated on labels that the data pipeline never attaches,
contributes plain-float zeros and is skipped by the total at :430-437; it is omitted.
"""
import torch
import torch.nn.functional as F

from .schedule import edm_loss_weight, edm_scalings, edm_sigma_from_u
from .score_network import score_network_forward

LOSS_WEIGHTS = {                      # train.py:66-90 (entries that can be tensors)
    "edm_coordinate": 1.0, "edm_preconditioning": 0.1,
    "pilot_property": 0.2, "pilot_guidance": 0.15, "pilot_null_conditioning": 0.05,
    "dit_attention": 0.1, "dit_adaptive_norm": 0.05,
    "e3_coordinate": 1.0, "e3_rotation": 0.1, "e3_translation": 0.05,
}


def _mean_pool(v, batch, B):
    s = torch.zeros(B, v.shape[1], dtype=v.dtype).index_add(0, batch, v)
    c = torch.bincount(batch, minlength=B).clamp(min=1).to(v.dtype)
    return s / c.unsqueeze(-1)


def batch_properties(properties, B):
    """train.py:327-337: what the loss step makes of ``batch.properties``.  Returns (properties [B, w], pilot_loss):
    a 1-D tensor becomes one row repeated for every molecule, a tensor with another row count is cut to B rows.
    The reference's own loader stores 14 values per molecule as a 1-D vector (data_utils.py:904-914), which PyG's
    collate concatenates to [14 B]: every molecule is then conditioned on that row (the model reads its first
    eight columns, model1.py:342-345) and F.mse_loss(prop_pred [B,15], properties [B,w]) raises for w != 15 inside
    the try block of train.py:388-397, so the PILOT family contributes nothing: pilot_loss = False."""
    if properties.dim() == 1:
        properties = properties.unsqueeze(0)
    if properties.shape[0] != B:
        properties = properties.repeat(B, 1) if properties.shape[0] == 1 else properties[:B]
    return properties, properties.shape[1] == 15


def loss_step(sd, x, pos, edge_index, batch, u, noise, properties=None, null_rand=None,
              training=True, max_atoms=100, sigma_min=0.002, sigma_max=80.0, sigma_data=0.5, shard=None,
              dropout=None):
    """Returns (losses dict of 0-dim tensors, aux dict).  B >= 2 (see edm_scalings).

    shard=(n_atoms_global, n_mols_global, rot_mols): evaluate ONE data-parallel shard's
    share of the global-batch loss - every mean divides by the global count and the
    first-two-molecules distance term only covers this shard's rot_mols leading molecules -
    so that the shard losses (and their gradients) of all ranks sum to the unsharded
    result.  Used by the tests of the multi-GPU decomposition; not a reference feature.
    dropout: the attention layers' train-mode masks (see score_network_forward)."""
    if shard is not None:
        return _loss_step_shard(sd, x, pos, edge_index, batch, u, noise, properties, null_rand, training,
                                max_atoms, sigma_min, sigma_max, sigma_data, shard)
    B = int(batch.max().item()) + 1
    sigma = edm_sigma_from_u(u, sigma_min, sigma_max)
    c_skip, c_out, c_in, c_noise = edm_scalings(sigma, sigma_data)
    w = edm_loss_weight(sigma, sigma_data)
    noisy = pos + sigma[batch].unsqueeze(-1) * noise                     # train.py:305-307
    pre = noisy * c_in[batch].unsqueeze(-1)                              # train.py:311-312
    props, pilot_loss = None, True
    if properties is not None:
        props, pilot_loss = batch_properties(properties, B)              # train.py:327-337
        if training:                                                     # train.py:340-343
            props = props.clone()
            props[null_rand < 0.1] = 0.0
    atom_pred, pos_pred, prop_pred = score_network_forward(
        sd, x, pre, edge_index, batch, c_noise, props, max_atoms=max_atoms, pad_to_max_atoms=False, dropout=dropout)
    L, den = loss_terms(atom_pred, pos_pred, prop_pred, pos, noise, noisy, c_skip, c_out, w, batch,
                        props if pilot_loss else None)
    aux = {"sigma": sigma, "denoised_pos": den, "atom_pred": atom_pred, "pos_pred": pos_pred,
           "prop_pred": prop_pred}
    return L, aux


def loss_terms(atom_pred, pos_pred, prop_pred, pos, noise, noisy, c_skip, c_out, w, batch, props):
    """Everything of train.py:358-437 that follows the model call: denoised coordinates, the loss
    families (train.py:454-473, :475-498, :603-623, :625-663), the weighted total with its NaN / Inf
    skip rule (train.py:430-437) and the family totals.  Returns (losses dict, denoised_pos)."""
    B = int(batch.max().item()) + 1
    den = c_skip[batch].unsqueeze(-1) * noisy + c_out[batch].unsqueeze(-1) * pos_pred  # :360-362
    L = {}
    L["edm_coordinate"] = (F.mse_loss(den, pos, reduction="none") * w[batch].unsqueeze(-1)).mean()
    L["edm_preconditioning"] = F.mse_loss(den - pos, noise) * 0.1
    if props is not None:
        L["pilot_property"] = F.mse_loss(prop_pred, props) * 0.2         # train.py:487-488
        L["pilot_guidance"] = torch.tensor(0.0)
        L["pilot_null_conditioning"] = torch.tensor(0.0)
    L["dit_attention"] = torch.var(atom_pred, dim=-1).mean() * 0.1       # train.py:614
    L["dit_adaptive_norm"] = F.mse_loss(pos_pred.norm(dim=-1), torch.ones(pos_pred.shape[0])) * 0.01
    L["e3_coordinate"] = F.mse_loss(den, pos)
    if B > 1:                                                            # train.py:640-654
        rot = 0.0
        for b in range(min(2, B)):
            m = batch == b
            if int(m.sum()) > 1:
                rot = rot + F.mse_loss(torch.cdist(den[m], den[m]), torch.cdist(pos[m], pos[m])) * 0.1
        if torch.is_tensor(rot):
            L["e3_rotation"] = rot
        L["e3_translation"] = F.mse_loss(_mean_pool(den, batch, B) - _mean_pool(pos, batch, B),
                                         torch.zeros(B, 3)) * 0.01
    total = torch.tensor(0.0)
    for k, wt in LOSS_WEIGHTS.items():                                   # train.py:430-437
        if k in L and not torch.isnan(L[k]) and not torch.isinf(L[k]):
            total = total + wt * L[k]
    L["edm_total"] = L["edm_coordinate"] + L["edm_preconditioning"]
    L["pilot_total"] = sum((L[k] for k in L if k.startswith("pilot_") and k != "pilot_total"), torch.tensor(0.0))
    L["dit_total"] = L["dit_attention"] + L["dit_adaptive_norm"]
    L["e3_total"] = sum((L[k] for k in ("e3_coordinate", "e3_rotation", "e3_translation") if k in L), torch.tensor(0.0))
    L["total_loss"] = total
    return L, den


def _loss_step_shard(sd, x, pos, edge_index, batch, u, noise, properties, null_rand, training, max_atoms,
                     sigma_min, sigma_max, sigma_data, shard):
    n_atoms_g, n_mols_g, rot_mols = shard
    B = int(batch.max().item()) + 1
    sigma = edm_sigma_from_u(u, sigma_min, sigma_max)
    # element-wise form of model1.py:27-34 (a shard may hold a single molecule)
    c_skip = sigma_data ** 2 / (sigma ** 2 + sigma_data ** 2)
    c_out = sigma * sigma_data / (sigma ** 2 + sigma_data ** 2).sqrt()
    c_in = 1 / (sigma_data ** 2 + sigma ** 2).sqrt()
    c_noise = sigma.log() / 4
    w = edm_loss_weight(sigma, sigma_data)
    noisy = pos + sigma[batch].unsqueeze(-1) * noise
    pre = noisy * c_in[batch].unsqueeze(-1)
    props = None
    if properties is not None:
        props = properties.clone()
        if training:
            props[null_rand < 0.1] = 0.0
    atom_pred, pos_pred, prop_pred = score_network_forward(
        sd, x, pre, edge_index, batch, c_noise, props, max_atoms=max_atoms, pad_to_max_atoms=False)
    den = c_skip[batch].unsqueeze(-1) * noisy + c_out[batch].unsqueeze(-1) * pos_pred
    L = {}
    d = den - pos
    L["edm_coordinate"] = (d ** 2 * w[batch].unsqueeze(-1)).sum() / (3 * n_atoms_g)
    L["edm_preconditioning"] = ((d - noise) ** 2).sum() / (3 * n_atoms_g) * 0.1
    if props is not None:
        L["pilot_property"] = ((prop_pred - props) ** 2).sum() / (15 * n_mols_g) * 0.2
    L["dit_attention"] = torch.var(atom_pred, dim=-1).sum() / n_atoms_g * 0.1
    L["dit_adaptive_norm"] = ((pos_pred.norm(dim=-1) - 1) ** 2).sum() / n_atoms_g * 0.01
    L["e3_coordinate"] = (d ** 2).sum() / (3 * n_atoms_g)
    if n_mols_g > 1:
        rot = torch.tensor(0.0)
        for b in range(rot_mols):
            m = batch == b
            if int(m.sum()) > 1:
                rot = rot + F.mse_loss(torch.cdist(den[m], den[m]), torch.cdist(pos[m], pos[m])) * 0.1
        L["e3_rotation"] = rot
        L["e3_translation"] = (_mean_pool(d, batch, B) ** 2).sum() / (3 * n_mols_g) * 0.01
    total = torch.tensor(0.0)
    for k, wt in LOSS_WEIGHTS.items():
        if k in L:
            total = total + wt * L[k]
    L["total_loss"] = total
    return L, {"denoised_pos": den}
