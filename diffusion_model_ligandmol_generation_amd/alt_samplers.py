"""The reference's two alternate samplers, batched over molecules on the HIP path.

* ``ddim_positions``  - ``ResearchAccurateDiffusionModel.sample`` (model1.py:890-966): chain
  connectivity, ``num_steps`` timesteps ``linspace(T-1, 0)``, classifier-free guidance when a
  target is given and the scale exceeds 1, and a coordinate-only update.
* ``euler_positions`` - ``generate_research_validated_molecules`` of the trainer
  (train.py:786-863): chain + six-ring connectivity on 25 atoms, the EDM sigma schedule
  (rho = 7) with sigma_t passed to the model as its time argument, Euler coordinate update.

The reference runs one molecule per forward; here ``num_samples`` equal-size molecules share every
forward (they are independent, so results per molecule are the same).  Both stop where the
reference hands over to RDKit (``_postprocess_molecule``, ``_postprocess_to_valid_molecule``):
that part stays on the host and is not rebuilt here.  Gaussian draws come from the library's
Philox stream (keyed by seed, step and global atom index) unless a ``noise`` callable supplies
them (parity runs).
"""
import torch

from .engine import PackedGraph


def chain_edges(num_atoms, ring=False):
    """model1.py:917-920; ring=True: train.py:894-914."""
    e = []
    for i in range(num_atoms - 1):
        e.extend([[i, i + 1], [i + 1, i]])
    if ring and num_atoms >= 6:
        for i in range(6):
            j = (i + 1) % 6
            if [i, j] not in e:
                e.extend([[i, j], [j, i]])
    if ring and not e:
        e = [[0, 1], [1, 0]]
    return e


def _batched_graph(num_atoms, num_samples, ring, device):
    if num_atoms < 2:
        raise ValueError("a one-atom molecule has no bonds")
    ei = torch.tensor(chain_edges(num_atoms, ring), dtype=torch.long, device=device).t().contiguous()
    off = (torch.arange(num_samples, device=device) * num_atoms).view(-1, 1, 1)
    ei = (ei.unsqueeze(0) + off).permute(1, 0, 2).reshape(2, -1).contiguous()
    batch = torch.arange(num_samples, device=device).repeat_interleave(num_atoms)
    return PackedGraph(ei, batch)


def _props(target_properties, num_samples, device):
    if target_properties is None:
        return None
    p = torch.as_tensor(target_properties, dtype=torch.float32, device=device)
    if p.dim() == 1:
        p = p.unsqueeze(0)
    if p.shape[0] == 1 and num_samples > 1:
        p = p.repeat(num_samples, 1)
    return p.contiguous()


def _predict(eng, graph, x, pos, t, props, guided):
    """(atom_pred, pos_pred, atom_pred_uncond, pos_pred_uncond); the last two None when unguided."""
    if guided:
        au, pu, _ = eng.forward(graph, x, pos, t, None, want_props=False)
        a, pc, _ = eng.forward(graph, x, pos, t, props, want_props=False)
        return a, pc, au, pu
    a, p, _ = eng.forward(graph, x, pos, t, props, want_props=False)
    return a, p, None, None


def ddim_positions(model, num_samples=1, target_properties=None, guidance_scale=1.0, num_steps=100, max_atoms=30,
                   seed=0, noise=None, init=None):
    """model1.py:890-966.  Returns (pos [S*max_atoms,3], atom_pred [S*max_atoms,119] of the last
    step, guidance-combined as at :949).  ``init`` = (x0, pos0) replaces the initial draws;
    ``noise(step_index, shape)`` replaces the per-step coordinate noise."""
    eng, dev = model.engine(), next(model.parameters()).device
    S, n = int(num_samples), int(max_atoms)
    graph = _batched_graph(n, S, False, dev)
    N = S * n
    if init is None:
        x = eng.philox_normal(N * 119, seed, 0xFFFFFFFF, 0).view(N, 119)
        pos = eng.philox_normal(N * 3, seed, 0xFFFFFFFF, 1).view(N, 3)
    else:
        x, pos = init[0].to(dev).float().contiguous(), init[1].to(dev).float().clone().contiguous()
    props = _props(target_properties, S, dev)
    guided = guidance_scale > 1.0 and props is not None                       # model1.py:937
    ts = torch.linspace(model.timesteps - 1, 0, num_steps, dtype=torch.long)  # model1.py:932
    ac = model.alphas_cumprod.detach().cpu()
    atom_pred = None
    with torch.no_grad():
        for i, t_val in enumerate(ts.tolist()):
            t = torch.full((1,), float(t_val), device=dev)
            a, pc, au, pu = _predict(eng, graph, x, pos, t, props, guided)
            atom_pred = au + guidance_scale * (a - au) if guided else a          # model1.py:949
            if i < len(ts) - 1:
                a_t, a_prev = ac[t_val], ac[ts[i + 1]]
                sigma_t = ((1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)).sqrt()    # model1.py:960
                noisy = bool(ts[i + 1] > 0)
                nz = noise(i, (N, 3)).to(dev).float().contiguous() if (noise is not None and noisy) else None
                eng.pos_step(pos, pc, float((1 - a_t).sqrt()), float(a_t.sqrt()), float(sigma_t) if noisy else 0.0,
                             pu, guidance_scale, nz, seed, i)
    return pos, atom_pred


def edm_sigma_steps(model, num_steps):
    """train.py:823-831 (host, fp32 as the reference computes it)."""
    edm = model.edm_preconditioning
    lo, hi, rho = edm.sigma_min, edm.sigma_max, edm.rho
    return (hi ** (1 / rho) + torch.linspace(0, 1, num_steps + 1) * (lo ** (1 / rho) - hi ** (1 / rho))) ** rho


def euler_positions(model, num_samples=10, target_properties=None, guidance_scale=7.5, num_steps=50, temperature=1.0,
                    num_atoms=25, seed=0, noise=None, init=None):
    """train.py:786-863.  Returns pos [S*num_atoms, 3]."""
    eng, dev = model.engine(), next(model.parameters()).device
    S, n = int(num_samples), int(num_atoms)
    graph = _batched_graph(n, S, True, dev)
    N = S * n
    if init is None:
        x = eng.philox_normal(N * 119, seed, 0xFFFFFFFF, 0).view(N, 119)
        pos = eng.philox_normal(N * 3, seed, 0xFFFFFFFF, 1, scale=2.0).view(N, 3)   # train.py:811
    else:
        x, pos = init[0].to(dev).float().contiguous(), init[1].to(dev).float().clone().contiguous()
    props = _props(target_properties, S, dev)
    guided = guidance_scale > 1.0 and props is not None                       # train.py:841
    sig = edm_sigma_steps(model, num_steps)
    with torch.no_grad():
        for step in range(num_steps):
            s_t, s_n = sig[step], sig[step + 1]
            t = torch.full((1,), float(s_t), device=dev)
            _, pc, _, pu = _predict(eng, graph, x, pos, t, props, guided)
            if step < num_steps - 1:
                noisy = temperature > 0 and float(s_n) > 0                    # train.py:861
                nz = noise(step, (N, 3)).to(dev).float().contiguous() if (noise is not None and noisy) else None
                c = float(s_n) * float(temperature) * 0.1 if noisy else 0.0
                eng.pos_step(pos, pc, float(s_t - s_n), 1.0, c, pu, guidance_scale, nz, seed, step)
    return pos
