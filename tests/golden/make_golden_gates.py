"""Writes tests/golden/gates.npz by RUNNING THE REFERENCE's generation.py in this container
(through tests/golden/_ref_loader.py; needs /root/reference, which the GPU box does not have).

Run from the repo root:  python tests/golden/make_golden_gates.py

Contents (inputs + the reference's own answers):
* validity gate: crafted molecules, including pairs just either side of every threshold of
  generation.py:162-233 (softmax max-probability 0.1, per-axis variance 1e-6 and 100, pairwise distance
  0.5 and 20), NaN / Inf entries, a row-count mismatch, one- and two-atom molecules
  -> ResearchValidatedGenerator._validate_research_compliance / _validate_final_molecule;
* prediction gate: prediction pairs around |value| = 100, NaN, Inf
  -> ResearchValidatedGenerator._validate_model_predictions (generation.py:518-541);
* posterior_variance: the buffer _add_missing_attributes_to_model registers (generation.py:671-680)
  for T = 1000 and T = 50;
* molecule sizes: _sample_realistic_molecule_size sequences under np.random.seed for four atom ranges
  (generation.py:126-160);
* outer loop: generate_with_research_protocols (generation.py:38-124) with the per-molecule sampler
  replaced by a stub that returns the crafted molecules above (a random-init model never passes the
  gates), so that the size sequence, the acceptance decisions and the result dictionaries are the
  reference's.
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


def grid(n, spacing=1.5):
    """n points of a cubic lattice."""
    side = int(np.ceil(n ** (1 / 3)))
    pts = [[i, j, k] for i in range(side) for j in range(side) for k in range(side)][:n]
    return torch.tensor(pts, dtype=torch.float32) * spacing


def dominant(n, logit=10.0, col=6):
    x = torch.zeros(n, 119)
    x[:, col] = logit
    return x


def validity_cases():
    g = torch.Generator().manual_seed(11)
    cases = []   # (tag, x, pos, num_atoms)

    def add(tag, x, pos, n=None):
        cases.append((tag, x.clone().float(), pos.clone().float(), x.shape[0] if n is None else n))

    add("pass_grid8", dominant(8), grid(8))
    add("pass_grid27", dominant(27), grid(27))
    add("random_normal_12", torch.randn(12, 119, generator=g), torch.randn(12, 3, generator=g) * 3)
    add("random_dominant_20", torch.randn(20, 119, generator=g) + dominant(20, 12.0), grid(20) + torch.randn(20, 3, generator=g) * 0.1)
    # softmax max-probability threshold 0.1: one logit a, 118 zeros -> p = e^a / (e^a + 118); p = 0.1 at a = 2.5735
    for a, tag in ((2.52, "maxprob_below"), (2.63, "maxprob_above")):
        x = dominant(8)
        x[4] = 0.0
        x[4, 17] = a
        add(tag, x, grid(8))
    # per-axis variance lower threshold 1e-6: z coordinates +-c alternate, unbiased variance c^2 n / (n - 1)
    for c, tag in ((0.85e-3, "var_low_below"), (1.1e-3, "var_low_above")):
        pos = torch.cat([torch.arange(16).float().view(-1, 1) % 4 * 1.5,
                         torch.arange(16).float().view(-1, 1) // 4 * 1.5,
                         (torch.arange(16).float().view(-1, 1) % 2 * 2 - 1) * c], dim=1)
        add(tag, dominant(16), pos)
    # per-axis variance upper threshold 100 with all distances <= 20: two 2 x 2 clusters at x = +-a (n = 8)
    for a, tag in ((9.30, "var_high_below"), (9.42, "var_high_above")):
        yz = torch.tensor([[0, 0], [0.6, 0], [0, 0.6], [0.6, 0.6]], dtype=torch.float32)
        pos = torch.cat([torch.cat([torch.full((4, 1), -a), yz], 1), torch.cat([torch.full((4, 1), a), yz], 1)])
        add(tag, dominant(8), pos)
    # minimum distance 0.5
    for d, tag in ((0.49, "dmin_below"), (0.51, "dmin_above")):
        pos = grid(8)
        pos[1] = pos[0] + torch.tensor([d, 0.0, 0.0])
        add(tag, dominant(8), pos)
    # maximum distance 20 with every per-axis variance <= 100: two atoms on the space diagonal, six near the centre
    for d, tag in ((19.8, "dmax_below"), (20.2, "dmax_above")):
        h = d / 2 / np.sqrt(3.0)
        pos = torch.cat([torch.tensor([[h, h, h], [-h, -h, -h]], dtype=torch.float32), grid(6, 0.8) - 0.4])
        add(tag, dominant(8), pos)
    for where, val, tag in (("x", float("nan"), "nan_x"), ("x", float("inf"), "inf_x"),
                            ("pos", float("nan"), "nan_pos"), ("pos", float("-inf"), "ninf_pos")):
        x, pos = dominant(8), grid(8)
        (x if where == "x" else pos)[3, 1] = val
        add(tag, x, pos)
    add("row_count_mismatch", dominant(8), grid(8), n=9)
    add("one_atom", dominant(1), grid(1))
    add("two_atoms_ok", dominant(2), torch.tensor([[0.0, 0, 0], [1.2, 0.9, 0.7]]))
    add("two_atoms_same_x", dominant(2), torch.tensor([[0.0, 0, 0], [0.0, 0.9, 0.7]]))
    add("all_same_position", dominant(5), torch.zeros(5, 3))
    add("diffuse_types", torch.zeros(8, 119), grid(8))
    add("large_128", dominant(128), grid(128, 1.2))
    add("spread_out", dominant(8), grid(8, 25.0))
    add("big_logits", dominant(8, 80.0), grid(8))
    add("negative_logits", -dominant(8, 30.0), grid(8))
    return cases


def prediction_cases():
    g = torch.Generator().manual_seed(12)
    out = []
    for tag, where, val in (("ok", None, 0.0), ("x_99.9", "x", 99.9), ("x_100", "x", 100.0), ("x_100.1", "x", 100.1),
                            ("x_-100.1", "x", -100.1), ("pos_99.9", "pos", 99.9), ("pos_100", "pos", 100.0),
                            ("pos_100.1", "pos", 100.1), ("pos_-150", "pos", -150.0), ("x_nan", "x", float("nan")),
                            ("x_inf", "x", float("inf")), ("pos_nan", "pos", float("nan")),
                            ("pos_-inf", "pos", float("-inf"))):
        n = 5 + len(out)
        ex, ep = torch.randn(n, 119, generator=g), torch.randn(n, 3, generator=g)
        if where == "x":
            ex[n // 2, 7] = val
        elif where == "pos":
            ep[n // 2, 1] = val
        out.append((tag, ex, ep))
    return out


def main():
    m1, tr, gen = _ref_loader.load_reference()
    G = gen.ResearchValidatedGenerator
    out = {}

    def quiet(fn):
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            return fn()

    torch.manual_seed(0)
    model = quiet(lambda: m1.ResearchAccurateDiffusionModel(timesteps=1000)).eval()
    generator = quiet(lambda: G(model))

    # ---- validity gate
    vc = validity_cases()
    out["valid_tags"] = np.array([c[0] for c in vc])
    out["valid_ptr"] = np.cumsum([0] + [c[1].shape[0] for c in vc]).astype(np.int64)
    out["valid_x"] = torch.cat([c[1] for c in vc]).numpy()
    out["valid_pos"] = torch.cat([c[2] for c in vc]).numpy()
    out["valid_num_atoms"] = np.array([c[3] for c in vc], dtype=np.int64)
    out["valid_compliance"] = np.array([bool(quiet(lambda c=c: generator._validate_research_compliance(c[1], c[2], c[3])))
                                        for c in vc])
    out["valid_final"] = np.array([bool(quiet(lambda c=c: generator._validate_final_molecule(c[1], c[2], c[3])))
                                   for c in vc])
    for tag, ok in zip(out["valid_tags"], out["valid_compliance"]):
        print(f"  validity {tag:22s} -> {bool(ok)}")
    want_pairs = {"maxprob": (False, True), "var_low": (False, True), "var_high": (True, False),
                  "dmin": (False, True), "dmax": (True, False)}
    res = dict(zip(out["valid_tags"].tolist(), out["valid_compliance"].tolist()))
    for k, (below, above) in want_pairs.items():     # each threshold pair must really straddle its threshold
        assert res[k + "_below"] == below and res[k + "_above"] == above, k

    # ---- prediction gate
    pc = prediction_cases()
    out["pred_tags"] = np.array([c[0] for c in pc])
    out["pred_ptr"] = np.cumsum([0] + [c[1].shape[0] for c in pc]).astype(np.int64)
    out["pred_x"] = torch.cat([c[1] for c in pc]).numpy()
    out["pred_pos"] = torch.cat([c[2] for c in pc]).numpy()
    out["pred_ok"] = np.array([bool(generator._validate_model_predictions(c[1], c[2])) for c in pc])
    for tag, ok in zip(out["pred_tags"], out["pred_ok"]):
        print(f"  prediction {tag:10s} -> {bool(ok)}")

    # ---- posterior_variance buffer
    assert hasattr(model, "posterior_variance")
    out["posterior_variance_T1000"] = model.posterior_variance.detach().numpy()
    torch.manual_seed(0)
    m50 = quiet(lambda: m1.ResearchAccurateDiffusionModel(timesteps=50)).eval()
    quiet(lambda: G(m50))
    out["posterior_variance_T50"] = m50.posterior_variance.detach().numpy()

    # ---- molecule size sequences
    ranges = [(10, 40), (10, 25), (25, 40), (5, 12)]
    out["size_ranges"] = np.array(ranges, dtype=np.int64)
    for i, r in enumerate(ranges):
        np.random.seed(7 + i)
        out[f"size_seq_{i}"] = np.array([int(generator._sample_realistic_molecule_size(r)) for _ in range(64)], dtype=np.int64)
    out["size_probs_15_31"] = np.array(generator._get_drug_size_probabilities(15, 31), dtype=np.float64)

    # ---- outer loop with a stub sampler: call k returns crafted case k of the list below (None = sampler gave up)
    torch.manual_seed(0)
    m5 = quiet(lambda: m1.ResearchAccurateDiffusionModel(timesteps=5)).eval()
    g5 = quiet(lambda: G(m5))
    calls = []

    def stub(num_atoms, target_properties, *a, **k):
        i = len(calls)
        kind = ("ok", "none", "overlap", "ok", "diffuse", "ok", "nan", "ok")[i % 8]
        x, pos = dominant(num_atoms, 9.0 + i, col=(6 + i) % 119), grid(num_atoms) + 0.01 * i
        if kind == "overlap":
            pos[1] = pos[0]
        elif kind == "diffuse":
            x = torch.zeros(num_atoms, 119)
        elif kind == "nan":
            x[0, 0] = float("nan")
        calls.append((kind, num_atoms, x.clone(), pos.clone()))
        return None if kind == "none" else (x, pos)

    g5.ddpm_sampling_fixed = stub
    g5.ddim_sampling_fixed = lambda n, tp, steps, gs, temp: stub(n, tp)
    tp = torch.linspace(0.1, 0.9, 15)
    np.random.seed(3)
    mols = quiet(lambda: g5.generate_with_research_protocols(num_molecules=12, target_properties=tp, atom_range=(10, 40),
                                                              guidance_scale=2.0, num_sampling_steps=4, temperature=1.0,
                                                              use_ddim=False))
    out["loop_kinds"] = np.array([c[0] for c in calls])
    out["loop_sizes"] = np.array([c[1] for c in calls], dtype=np.int64)
    out["loop_ptr"] = np.cumsum([0] + [c[1] for c in calls]).astype(np.int64)
    out["loop_x"] = torch.cat([c[2] for c in calls]).numpy()
    out["loop_pos"] = torch.cat([c[3] for c in calls]).numpy()
    out["loop_target_properties"] = tp.numpy()
    out["loop_result_ids"] = np.array([m["generation_id"] for m in mols], dtype=np.int64)
    out["loop_result_num_atoms"] = np.array([m["num_atoms"] for m in mols], dtype=np.int64)
    out["loop_result_keys"] = np.array(sorted(mols[0].keys()))
    for m in mols:
        i = m["generation_id"]
        assert m["atom_features"].dtype == np.float32 and m["positions"].dtype == np.float32
        assert np.array_equal(m["atom_features"], calls[i][2].numpy()) and np.array_equal(m["positions"], calls[i][3].numpy())
        assert np.array_equal(m["target_properties"], tp.numpy())
        assert m["generated_by_model"] is True and m["research_compliant"] is True
    print("  outer loop: kept generation ids", out["loop_result_ids"].tolist(), "sizes", out["loop_sizes"].tolist())
    np.savez_compressed(os.path.join(HERE, "gates.npz"), **out)
    print("written", os.path.join(HERE, "gates.npz"))


if __name__ == "__main__":
    main()
