"""Oracle (test infrastructure): DDPM / DDIM reverse loops and the validity gates.

One molecule at a time, as generation.py runs it.  All Gaussian draws come from a
caller-supplied ``noise(shape) -> tensor`` so that a parity test can feed the HIP path
the very same numbers.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .connectivity import template_connectivity
from .score_network import score_network_forward


def ddpm_step(x, eps, t, sched, z=None, temperature=1.0):
    """One reverse update for either state tensor, generation.py:457-484.

    t > 0: x <- (x - beta_t / sqrt(1 - abar_t) * eps) / sqrt(alpha_t), then for t > 1
           x <- x + z * sqrt(beta_t) * temperature (the noise scale is sqrt(beta_t), not the
           posterior variance).
    t = 0: x <- (x - sqrt(1 - abar_0) * eps) / sqrt(abar_0).
    """
    if t > 0:
        beta, alpha, abar = sched["betas"][t], sched["alphas"][t], sched["alphas_cumprod"][t]
        x = (1 / torch.sqrt(alpha)) * (x - (beta / torch.sqrt(1 - abar)) * eps)
        if t > 1:
            x = x + z * torch.sqrt(beta) * temperature
        return x
    abar = sched["alphas_cumprod"][t]
    return (x - torch.sqrt(1 - abar) * eps) / torch.sqrt(abar)


def _predict(sd, x, pos, edge_index, batch, t, props, guidance_scale, max_atoms, pad):
    """generation.py:428-451: classifier-free combination when props given and scale > 1.
    The 'unconditional' branch passes properties=None (no property term at all)."""
    tb = torch.tensor([t])
    if props is not None and not torch.is_tensor(props):
        # dict-valued properties (the scenarios of generation.py:689-757 reach the model unchanged, :510-512): the
        # model's forward fails on them (model1.py:337, no .shape) and returns all-zero predictions (model1.py:757-764)
        if guidance_scale > 1.0:
            au, pu, _ = score_network_forward(sd, x, pos, edge_index, batch, tb, None,
                                              max_atoms=max_atoms, pad_to_max_atoms=pad)
            return au + guidance_scale * (0.0 - au), pu + guidance_scale * (0.0 - pu)
        return torch.zeros_like(x), torch.zeros_like(pos)
    if props is not None and guidance_scale > 1.0:
        ac, pc, _ = score_network_forward(sd, x, pos, edge_index, batch, tb, props,
                                          max_atoms=max_atoms, pad_to_max_atoms=pad)
        au, pu, _ = score_network_forward(sd, x, pos, edge_index, batch, tb, None,
                                          max_atoms=max_atoms, pad_to_max_atoms=pad)
        return au + guidance_scale * (ac - au), pu + guidance_scale * (pc - pu)
    a, p, _ = score_network_forward(sd, x, pos, edge_index, batch, tb, props,
                                    max_atoms=max_atoms, pad_to_max_atoms=pad)
    return a, p


def _edges(num_atoms, edge_index=None):
    """The generator's bond template (generation.py:409-414), or the caller's [2,E] edge_index of
    the one molecule (any graph the model accepts, model1.py:723)."""
    if edge_index is not None:
        ei = torch.as_tensor(edge_index, dtype=torch.long)
        if ei.dim() != 2 or ei.shape[0] != 2:
            raise ValueError("edge_index must be [2, E]")
        return ei
    conn = template_connectivity(num_atoms)
    if conn is None:
        return None
    ei = torch.tensor(conn[0], dtype=torch.long).t()
    if ei.dim() != 2:
        raise ValueError("a molecule without bonds is outside the parity domain")
    return ei


def ddpm_sample(sd, sched, num_atoms, noise, target_properties=None, guidance_scale=1.0,
                temperature=1.0, timesteps=None, max_atoms=100, pad=True, trace=None,
                atom_feature_dim=119, edge_index=None):
    """generation.py:393-484 up to (not including) the final validity gate.  edge_index [2,E]
    replaces the bond template of generation.py:409-414 when given.

    noise(shape) is called in the reference's order: x0, pos0, then per step t > 1 the
    feature noise followed by the coordinate noise.  trace(t, x, pos), if given, sees the
    state that enters step t.  Returns (x, pos)."""
    T = sched["betas"].shape[0] if timesteps is None else timesteps
    x = noise((num_atoms, atom_feature_dim)) * temperature
    pos = noise((num_atoms, 3)) * temperature
    ei = _edges(num_atoms, edge_index)
    batch = torch.zeros(num_atoms, dtype=torch.long)
    props = target_properties
    if torch.is_tensor(props) and props.dim() == 1:
        props = props.unsqueeze(0)                                  # generation.py:503-504
    for t in reversed(range(T)):
        if trace is not None:
            trace(t, x, pos)
        ex, ep = _predict(sd, x, pos, ei, batch, t, props, guidance_scale, max_atoms, pad)
        zx = noise(tuple(x.shape)) if t > 1 else None
        zp = noise(tuple(pos.shape)) if t > 1 else None
        x = ddpm_step(x, ex, t, sched, zx, temperature)
        pos = ddpm_step(pos, ep, t, sched, zp, temperature)
    return x, pos


def ddim_timesteps(T, num_steps):
    """generation.py:305: linspace(T-1, 0, num_steps) truncated to int64."""
    return torch.linspace(T - 1, 0, num_steps, dtype=torch.long)


def ddim_sample(sd, sched, num_atoms, noise, num_steps, target_properties=None,
                guidance_scale=1.0, temperature=1.0, max_atoms=100, pad=True, trace=None,
                atom_feature_dim=119, edge_index=None):
    """generation.py:279-381 up to the final gate (edge_index as in ddpm_sample).  Returns (x, pos), or None when the
    per-step prediction gate (generation.py:351, :518-541) rejects."""
    T = sched["betas"].shape[0]
    x = noise((num_atoms, atom_feature_dim)) * temperature
    pos = noise((num_atoms, 3)) * temperature
    ei = _edges(num_atoms, edge_index)
    batch = torch.zeros(num_atoms, dtype=torch.long)
    ts = ddim_timesteps(T, num_steps)
    props = target_properties
    if torch.is_tensor(props) and props.dim() == 1:
        props = props.unsqueeze(0)
    ac = sched["alphas_cumprod"]
    for i, t in enumerate(ts.tolist()):
        if trace is not None:
            trace(t, x, pos)
        ex, ep = _predict(sd, x, pos, ei, batch, t, props, guidance_scale, max_atoms, pad)
        if not prediction_gate(ex, ep):
            return None
        a_t = ac[t]
        x0 = (x - torch.sqrt(1 - a_t) * ex) / torch.sqrt(a_t)
        p0 = (pos - torch.sqrt(1 - a_t) * ep) / torch.sqrt(a_t)
        if i < len(ts) - 1:
            a_n = ac[ts[i + 1]]
            x = torch.sqrt(a_n) * x0 + torch.sqrt(1 - a_n) * ex
            pos = torch.sqrt(a_n) * p0 + torch.sqrt(1 - a_n) * ep
        else:
            x, pos = x0, p0
    return x, pos


def prediction_gate(ex, ep):
    """generation.py:518-541: finite and |.| <= 100."""
    for v in (ex, ep):
        if torch.isnan(v).any() or torch.isinf(v).any() or v.abs().max() > 100:
            return False
    return True


def validity_gate(x, pos, num_atoms):
    """generation.py:162-233 (also reached through :543-560).  Returns bool."""
    if x.shape[0] != num_atoms or pos.shape[0] != num_atoms:
        return False
    if x.shape[1] != 119 or pos.shape[1] != 3:
        return False
    for v in (x, pos):
        if torch.isnan(v).any() or torch.isinf(v).any():
            return False
    if torch.any(F.softmax(x, dim=-1).max(dim=-1)[0] < 0.1):
        return False
    var = torch.var(pos, dim=0)
    if torch.any(var < 1e-6) or torch.any(var > 100):
        return False
    if num_atoms > 1:
        d = torch.cdist(pos, pos)[~torch.eye(num_atoms, dtype=torch.bool)]
        if torch.any(d < 0.5) or torch.any(d > 20):
            return False
    return True


def sample_molecule_size(atom_range, rng=np.random):
    """generation.py:126-160."""
    lo, hi = atom_range
    if lo <= 20 <= hi:
        sizes = list(range(15, min(31, hi)))
        p = [0.15 if 18 <= s <= 25 else (0.1 if 15 <= s <= 30 else 0.05) for s in sizes]
        tot = sum(p)
        return int(rng.choice(sizes, p=[q / tot for q in p]))
    return int(rng.randint(lo, hi))


# ---------------------------------------------------------------------------------------
# The two alternate per-molecule samplers.  Both only move the coordinates; the atom
# features keep their initial Gaussian draw.

def chain_edges(num_atoms, ring=False):
    """Bond list of model1.py:917-920 (chain, both directions); with ring=True the trainer's
    variant, train.py:894-914, which also closes atoms 0..5 into a ring when n >= 6."""
    e = []
    for i in range(num_atoms - 1):
        e.extend([[i, i + 1], [i + 1, i]])
    if ring and num_atoms >= 6:
        for i in range(6):
            j = (i + 1) % 6
            if [i, j] not in e:
                e.extend([[i, j], [j, i]])
    if ring and len(e) == 0:
        e = [[0, 1], [1, 0]]
    return e


def _predict_t(sd, x, pos, ei, batch, t_tensor, props, guidance_scale, max_atoms):
    """model1.py:937-952 / train.py:841-853: unconditional + scale * (conditional - unconditional)
    when guided, else one forward with the properties as given."""
    if guidance_scale > 1.0 and props is not None:
        au, pu, _ = score_network_forward(sd, x, pos, ei, batch, t_tensor, None, max_atoms=max_atoms)
        ac, pc, _ = score_network_forward(sd, x, pos, ei, batch, t_tensor, props, max_atoms=max_atoms)
        return au + guidance_scale * (ac - au), pu + guidance_scale * (pc - pu)
    a, p, _ = score_network_forward(sd, x, pos, ei, batch, t_tensor, props, max_atoms=max_atoms)
    return a, p


def model_sample_positions(sd, sched, x0, pos0, noise, num_steps, target_properties=None,
                           guidance_scale=1.0, max_atoms=100):
    """model1.py:890-966 for ONE molecule up to (not including) the RDKit post-processing.

    x0 [n,119], pos0 [n,3]: the initial draws; noise(shape) supplies each step's coordinate
    noise in the reference's order.  Returns (pos, atom_pred of the last step)."""
    n = x0.shape[0]
    T = sched["betas"].shape[0]
    ei = torch.tensor(chain_edges(n), dtype=torch.long).t().contiguous()
    batch = torch.zeros(n, dtype=torch.long)
    ts = torch.linspace(T - 1, 0, num_steps, dtype=torch.long)
    ac = sched["alphas_cumprod"]
    x, pos = x0, pos0
    atom_pred = None
    for i, t_val in enumerate(ts):
        atom_pred, pos_pred = _predict_t(sd, x, pos, ei, batch, torch.tensor([int(t_val)]), target_properties,
                                         guidance_scale, max_atoms)
        if i < len(ts) - 1:
            a_t, a_prev = ac[t_val], ac[ts[i + 1]]
            sigma_t = ((1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)).sqrt()
            pos = (pos - pos_pred * (1 - a_t).sqrt()) / a_t.sqrt()
            if ts[i + 1] > 0:
                pos = pos + sigma_t * noise(tuple(pos.shape))
    return pos, atom_pred


def edm_sigma_steps(num_steps, sigma_min=0.002, sigma_max=80.0, rho=7.0):
    """train.py:829-831."""
    return (sigma_max ** (1 / rho) + torch.linspace(0, 1, num_steps + 1) *
            (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho


def trainer_euler_positions(sd, x0, pos0, noise, num_steps, target_properties=None, guidance_scale=7.5,
                            temperature=1.0, sigma_min=0.002, sigma_max=80.0, rho=7.0, max_atoms=100):
    """train.py:786-863 for ONE molecule up to the RDKit post-processing.  The model is
    called with sigma_t as its time argument.  Returns pos."""
    n = x0.shape[0]
    ei = torch.tensor(chain_edges(n, ring=True), dtype=torch.long).t().contiguous()
    batch = torch.zeros(n, dtype=torch.long)
    sig = edm_sigma_steps(num_steps, sigma_min, sigma_max, rho)
    x, pos = x0, pos0
    for step in range(num_steps):
        s_t, s_n = sig[step], sig[step + 1]
        _, pos_pred = _predict_t(sd, x, pos, ei, batch, s_t.unsqueeze(0), target_properties, guidance_scale, max_atoms)
        if step < num_steps - 1:
            pos = pos - (s_t - s_n) * pos_pred
            if temperature > 0 and s_n > 0:
                pos = pos + s_n * (noise(tuple(pos.shape)) * temperature) * 0.1
    return pos
