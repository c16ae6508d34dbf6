"""Writes the golden fixtures under tests/golden/ by RUNNING THE REFERENCE in this container.

Run from the repo root:  python tests/golden/make_golden.py
It needs /root/reference (read-only) and tests/golden/_ref_loader.py; neither exists on
the GPU box, where only the resulting .npz files are read.  A fixture holds inputs and
the reference's outputs.  The 6.6 M default-initialised weights are not stored: they are
a pure function of torch.manual_seed(0) and the module construction order, so a fixture
stores per-tensor checksums instead and the tests rebuild the weights from the seed and
verify the checksums before using them.
"""
import io
import contextlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_loader  # noqa: E402

OUT = HERE
torch.set_num_threads(8)


def checksums(sd):
    out = {}
    for k, v in sd.items():
        v = v.detach().double()
        out[k] = np.array([float(v.sum()), float(v.abs().sum()), float(v.flatten()[0]), float(v.flatten()[-1])])
    return out


def build_model(m1, seed=0, **kw):
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        model = m1.ResearchAccurateDiffusionModel(**kw)
    return model


def template_batch(gen_cls, sizes, g):
    """Ragged batch with the reference's own template connectivity per molecule."""
    xs, ps, es, bs = [], [], [], []
    off = 0
    for m, n in enumerate(sizes):
        conn = gen_cls.create_research_connectivity_fixed(None, n)
        ei = torch.tensor(conn[0], dtype=torch.long).t()
        xs.append(torch.randn(n, 119, generator=g))
        ps.append(torch.randn(n, 3, generator=g) * 1.5)
        es.append(ei + off)
        bs.append(torch.full((n,), m, dtype=torch.long))
        off += n
    return torch.cat(xs), torch.cat(ps), torch.cat(es, 1), torch.cat(bs)


def run_checked(fn):
    """Runs fn and fails if the reference took one of its print-and-swallow branches."""
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = fn()
    txt = buf.getvalue()
    for bad in ("failed", "error", "Failed", "Error"):
        assert bad not in txt, txt
    return out


def main():
    m1, tr, gen = _ref_loader.load_reference()
    Data = sys.modules["torch_geometric.data"].Data
    G = gen.ResearchValidatedGenerator

    # ---- 1. template connectivity (integer exact) --------------------------------
    conn = {}
    for n in list(range(1, 41)) + [48, 64, 65, 100, 127, 128]:
        c = G.create_research_connectivity_fixed(None, n)
        conn[f"edges_{n}"] = np.asarray(c[0], dtype=np.int64).reshape(-1, 2)
        conn[f"feats_{n}"] = np.asarray(c[1], dtype=np.float32).reshape(-1, 5)
    np.savez_compressed(os.path.join(OUT, "connectivity.npz"), **conn)

    # ---- 2. schedule buffers ------------------------------------------------------
    model = build_model(m1).eval()
    sd = model.state_dict()
    sched = {k: sd[k].numpy() for k in ("betas", "alphas", "alphas_cumprod",
                                         "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod")}
    model50 = build_model(m1, timesteps=50).eval()
    sched.update({k + "_T50": model50.state_dict()[k].numpy() for k in ("betas", "alphas", "alphas_cumprod")})
    np.savez_compressed(os.path.join(OUT, "schedule.npz"), **sched)

    ck = checksums(sd)
    np.savez_compressed(os.path.join(OUT, "weights_seed0_checksums.npz"), **ck)

    # ---- 3. eval forward ------------------------------------------------------------
    g = torch.Generator().manual_seed(1234)
    cases = {}
    sizes = [5, 12, 20, 29, 7, 16]
    x, pos, ei, batch = template_batch(G, sizes, g)
    B = len(sizes)
    props = torch.rand(B, 15, generator=g)
    t_int = torch.tensor([999, 500, 1, 0, 37, 742])
    t_flt = torch.randn(B, generator=g) * 1.2 - 0.3        # c_noise-like values
    for name, t, p, cfg in (("int_noprop", t_int, None, False), ("flt_noprop", t_flt, None, False),
                            ("int_prop", t_int, props, False), ("int_cfgnull", t_int, props, True)):
        with torch.no_grad():
            a, pp, pr, cons = run_checked(
                lambda: model(Data(x=x, pos=pos, edge_index=ei, batch=batch), t, p, cfg))
        assert cons is None and float(a.abs().sum()) > 0
        cases[f"{name}_atom"] = a.numpy()
        cases[f"{name}_pos"] = pp.numpy()
        cases[f"{name}_prop"] = pr.numpy()
    # larger model limit: fully connected 40-atom molecule + a 128-atom template one, max_atoms=128
    model128 = build_model(m1, max_atoms=128).eval()
    n_fc = 40
    ii, jj = torch.meshgrid(torch.arange(n_fc), torch.arange(n_fc), indexing="ij")
    fc = torch.stack([ii[ii != jj], jj[ii != jj]])
    c128 = G.create_research_connectivity_fixed(None, 128)
    e128 = torch.tensor(c128[0], dtype=torch.long).t() + n_fc
    x2 = torch.randn(n_fc + 128, 119, generator=g)
    pos2 = torch.randn(n_fc + 128, 3, generator=g) * 1.5
    ei2 = torch.cat([fc, e128], 1)
    b2 = torch.cat([torch.zeros(n_fc, dtype=torch.long), torch.ones(128, dtype=torch.long)])
    t2 = torch.tensor([3, 600])
    with torch.no_grad():
        a2, p2, r2, _ = run_checked(lambda: model128(Data(x=x2, pos=pos2, edge_index=ei2, batch=b2), t2, None))
    np.savez_compressed(
        os.path.join(OUT, "forward.npz"), sizes=np.array(sizes), x=x.numpy(), pos=pos.numpy(),
        edge_index=ei.numpy(), batch=batch.numpy(), props=props.numpy(), t_int=t_int.numpy(),
        t_flt=t_flt.numpy(), big_x=x2.numpy(), big_pos=pos2.numpy(), big_edge_index=ei2.numpy(),
        big_batch=b2.numpy(), big_t=t2.numpy(), big_atom=a2.numpy(), big_pos_pred=p2.numpy(),
        big_prop=r2.numpy(), **cases)

    # ---- 4. trainer loss step + gradients (dropout p = 0) --------------------------
    tmodel = build_model(m1)
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
    loss_out = {}
    for tag, with_props in (("noprop", False), ("prop", True)):
        batch_obj = Data(x=xt, pos=post, edge_index=eit, edge_attr=None, batch=bt)
        if with_props:
            batch_obj.properties = props_t.clone()
        seed = 777 if with_props else 555
        # replay the reference's draw order to record the numbers it will use
        torch.manual_seed(seed)
        u = torch.cat([torch.rand(1) for _ in range(Bt)])
        noise = torch.randn_like(post)
        null_rand = torch.rand(Bt) if with_props else torch.zeros(Bt)
        torch.manual_seed(seed)
        tmodel.zero_grad()
        losses = run_checked(lambda: trainer._compute_research_exact_loss(batch_obj, training=True))
        losses["total_loss"].backward()
        for k, v in losses.items():
            if torch.is_tensor(v):
                loss_out[f"{tag}_loss_{k}"] = np.float64(v.item())
        gsq = 0.0
        for name, p in tmodel.named_parameters():
            if p.grad is not None:
                gd = p.grad.double()
                gsq += float((gd ** 2).sum())
                loss_out[f"{tag}_gradsum_{name}"] = np.array(
                    [float(gd.sum()), float(gd.abs().sum()), float((gd ** 2).sum())])
                if p.numel() <= 4096:
                    loss_out[f"{tag}_grad_{name}"] = p.grad.numpy().copy()
        loss_out[f"{tag}_gradnorm"] = np.float64(gsq ** 0.5)
        loss_out[f"{tag}_u"] = u.numpy()
        loss_out[f"{tag}_noise"] = noise.numpy()
        loss_out[f"{tag}_null_rand"] = null_rand.numpy()
    np.savez_compressed(os.path.join(OUT, "loss_step.npz"), sizes=np.array(sizes_t), x=xt.numpy(),
                        pos=post.numpy(), edge_index=eit.numpy(), batch=bt.numpy(),
                        props=props_t.numpy(), **loss_out)

    # ---- 5. sampler trajectories ----------------------------------------------------
    samp = {}
    for tag, T, n, ddim_steps, with_props in (("ddpm_T50", 50, 12, None, False),
                                              ("ddpm_T1000", 1000, 20, None, False),
                                              ("ddpm_T50_cfg", 50, 9, None, True),
                                              ("ddim_T1000", 1000, 17, 20, False)):
        smodel = build_model(m1, timesteps=T).eval()
        with contextlib.redirect_stdout(io.StringIO()):
            generator = G(smodel)
        snaps = {}

        orig_forward = smodel.forward

        def spy(data, t, properties=None, classifier_free_guidance=False, _s=snaps, _f=orig_forward):
            ti = int(t.item())
            if properties is None or ti not in _s:
                pass
            if ti not in _s:
                _s[ti] = (data.x.clone(), data.pos.clone())
            return _f(data, t, properties, classifier_free_guidance)

        smodel.forward = spy
        final = {}

        def capture(x_, pos_, n_, _fin=final):
            _fin["x"], _fin["pos"] = x_.clone(), pos_.clone()
            return True

        generator._validate_final_molecule = capture
        seed = 4321
        tp = torch.rand(15, generator=g) if with_props else None
        torch.manual_seed(seed)
        with torch.no_grad():
            if ddim_steps is None:
                res = run_checked(lambda: generator.ddpm_sampling_fixed(n, tp, 2.0, 0.9))
            else:
                res = run_checked(lambda: generator.ddim_sampling_fixed(n, tp, ddim_steps, 2.0, 0.9))
        assert res is not None, tag
        keep = sorted(snaps) if T <= 50 or ddim_steps else [t for t in sorted(snaps) if t % 100 == 0 or t < 3 or t > T - 3]
        samp[f"{tag}_snap_t"] = np.array(keep)
        samp[f"{tag}_snap_x"] = np.stack([snaps[t][0].numpy() for t in keep])
        samp[f"{tag}_snap_pos"] = np.stack([snaps[t][1].numpy() for t in keep])
        samp[f"{tag}_final_x"] = final["x"].numpy()
        samp[f"{tag}_final_pos"] = final["pos"].numpy()
        samp[f"{tag}_meta"] = np.array([T, n, ddim_steps or 0, seed, int(with_props)])
        if tp is not None:
            samp[f"{tag}_props"] = tp.numpy()
    np.savez_compressed(os.path.join(OUT, "sampler.npz"), temperature=np.float32(0.9),
                        guidance_scale=np.float32(2.0), **samp)
    print("fixtures written to", OUT)


if __name__ == "__main__":
    main()
