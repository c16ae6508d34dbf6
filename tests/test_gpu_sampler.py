"""GPU: reverse-diffusion loops through the C-ABI against the reference's trajectories
(fixtures) and the oracle.  Per-step tolerance fp32 rtol 1e-4; free-running trajectories
are compared at a looser bound and their drift is printed."""
import copy

import numpy as np
import pytest
import torch

import oracle
from conftest import golden, seed0_model, seed0_state_dict

pytestmark = pytest.mark.gpu


def _gen(T=1000, max_atoms=100):
    from diffusion_model_ligandmol_generation_amd.generation import ResearchValidatedGenerator
    m = copy.deepcopy(seed0_model(timesteps=T, max_atoms=max_atoms)).cuda().eval()
    return ResearchValidatedGenerator(m)


class SeededNoise:
    """Replays torch.manual_seed(seed) CPU draws in the reference's call order
    (generation.py:405-406, :469-470): x0, pos0, then (x, pos) for t = T-1 ... 2."""

    def __init__(self, seed, n, T):
        g = torch.Generator().manual_seed(seed)
        self.d = {("x0", T): torch.randn(n, 119, generator=g), ("pos0", T): torch.randn(n, 3, generator=g)}
        for t in range(T - 1, 1, -1):
            self.d[("x", t)] = torch.randn(n, 119, generator=g)
            self.d[("pos", t)] = torch.randn(n, 3, generator=g)

    def __call__(self, kind, t, shape):
        return self.d[(kind, t)].cuda()


def test_template_edges_bit_exact():
    from diffusion_model_ligandmol_generation_amd.engine import template_graph
    f = golden("connectivity.npz")
    sizes = sorted(int(k.split("_")[1]) for k in f.files if k.startswith("edges_"))
    sizes = [n for n in sizes if n >= 2]
    graph, ei, kind = template_graph(sizes, torch.device("cuda"))
    ei, kind = ei.cpu().numpy(), kind.cpu().numpy()
    off_a = off_e = 0
    for n in sizes:
        want = f[f"edges_{n}"]
        got = ei[:, off_e:off_e + len(want)].T - off_a
        assert np.array_equal(got, want), n
        assert np.array_equal(kind[off_e:off_e + len(want)], (f[f"feats_{n}"][:, 3] == 1.0).astype(np.uint8)), n
        off_a += n
        off_e += len(want)
    assert off_e == ei.shape[1]


@pytest.mark.parametrize("tag", ["ddpm_T50", "ddpm_T50_cfg", "ddpm_T1000"])
def test_ddpm_trajectory_matches_reference(tag):
    s = golden("sampler.npz")
    T, n, _, seed, wp = [int(v) for v in s[tag + "_meta"]]
    gen = _gen(T)
    noise = SeededNoise(seed, n, T)
    tp = torch.from_numpy(s[tag + "_props"]) if wp else None
    snaps = {}
    x, pos, _, _ = gen.sample_batch([n], tp, 2.0, 0.9, noise=noise,
                                    trace=lambda t, a, b: snaps.__setitem__(t, (a.cpu().clone(), b.cpu().clone())))
    drift = 0.0
    for i, t in enumerate(s[tag + "_snap_t"]):
        rx, rp = torch.from_numpy(s[tag + "_snap_x"][i]), torch.from_numpy(s[tag + "_snap_pos"][i])
        drift = max(drift, float((snaps[int(t)][0] - rx).abs().max() / rx.abs().max()),
                    float((snaps[int(t)][1] - rp).abs().max() / rp.abs().max()))
    fx, fp = torch.from_numpy(s[tag + "_final_x"]), torch.from_numpy(s[tag + "_final_pos"])
    print(f"{tag}: max relative trajectory drift vs reference {drift:.3e}")
    torch.testing.assert_close(x.cpu(), fx, rtol=1e-3, atol=1e-3 * float(fx.abs().max()))
    torch.testing.assert_close(pos.cpu(), fp, rtol=1e-3, atol=1e-3 * float(fp.abs().max()))
    assert drift < 1e-3


def test_ddpm_single_steps_match_reference_teacher_forced():
    """One reverse step from the reference's own state at t must land on its state at t-1."""
    tag = "ddpm_T50"
    s = golden("sampler.npz")
    T, n, _, seed, _ = [int(v) for v in s[tag + "_meta"]]
    gen = _gen(T)
    model, eng = gen.model, gen.model.engine()
    from diffusion_model_ligandmol_generation_amd.engine import template_graph
    graph, _, _ = template_graph([n], torch.device("cuda"))
    noise = SeededNoise(seed, n, T)
    ts = {int(t): i for i, t in enumerate(s[tag + "_snap_t"])}
    t_index = torch.zeros(1, dtype=torch.int32, device="cuda")
    for t in (49, 30, 2, 1, 0):
        x = torch.from_numpy(s[tag + "_snap_x"][ts[t]]).cuda().clone()
        pos = torch.from_numpy(s[tag + "_snap_pos"][ts[t]]).cuda().clone()
        t_index.fill_(t)
        with torch.no_grad():
            ex, ep, _ = eng.forward(graph, x, pos, None, None, want_props=False, t_index=t_index)
        nx, npos = (noise("x", t, None), noise("pos", t, None)) if t > 1 else (None, None)
        eng.ddpm_step(eng.step_args(model, x, pos, ex, ep, temperature=0.9, t_index=t_index, noise_x=nx, noise_pos=npos))
        assert int(t_index.item()) == t - 1
        if t > 0:
            wx, wp = s[tag + "_snap_x"][ts[t - 1]], s[tag + "_snap_pos"][ts[t - 1]]
        else:
            wx, wp = s[tag + "_final_x"], s[tag + "_final_pos"]
        torch.testing.assert_close(x.cpu(), torch.from_numpy(wx), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(pos.cpu(), torch.from_numpy(wp), rtol=1e-4, atol=1e-4)


def test_ddim_trajectory_matches_reference():
    tag = "ddim_T1000"
    s = golden("sampler.npz")
    T, n, steps, seed, _ = [int(v) for v in s[tag + "_meta"]]
    gen = _gen(T)
    noise = SeededNoise(seed, n, 2)   # only the two initial draws are used
    noise.d[("x0", T)], noise.d[("pos0", T)] = noise.d.pop(("x0", 2)), noise.d.pop(("pos0", 2))
    snaps = {}
    x, pos, _, rejected = gen.sample_batch([n], None, 2.0, 0.9, use_ddim=True, num_steps=steps, noise=noise,
                                           trace=lambda t, a, b: snaps.__setitem__(t, (a.cpu().clone(), b.cpu().clone())))
    assert int(rejected[0]) == 0
    for i, t in enumerate(s[tag + "_snap_t"]):
        torch.testing.assert_close(snaps[int(t)][0], torch.from_numpy(s[tag + "_snap_x"][i]), rtol=1e-3, atol=1e-3)
    fx = torch.from_numpy(s[tag + "_final_x"])
    torch.testing.assert_close(x.cpu(), fx, rtol=1e-3, atol=1e-3 * float(fx.abs().max()))


def test_graph_replay_equals_host_loop_and_sharding_is_bitwise_invariant():
    gen = _gen(T=20)
    sizes = [9, 17, 29, 5]
    xa, pa, ga, _ = gen.sample_batch(sizes, None, 1.0, 1.0, seed=11, use_graph=True)
    xb, pb, _, _ = gen.sample_batch(sizes, None, 1.0, 1.0, seed=11, use_graph=False)
    assert torch.equal(xa, xb) and torch.equal(pa, pb)
    # the same molecules sampled as two shards (as two GPUs would) give the same bits
    x1, p1, _, _ = gen.sample_batch(sizes[:2], None, 1.0, 1.0, seed=11, atom_offset=0)
    x2, p2, _, _ = gen.sample_batch(sizes[2:], None, 1.0, 1.0, seed=11, atom_offset=sum(sizes[:2]))
    assert torch.equal(torch.cat([x1, x2]), xa) and torch.equal(torch.cat([p1, p2]), pa)
    assert torch.isfinite(xa).all() and torch.isfinite(pa).all()


@pytest.mark.parametrize("connectivity", ["template", "full"])
def test_two_concurrent_halves_equal_the_single_call_bitwise(connectivity):
    """Large DDPM batches run as two concurrent library calls on two streams (generation.py here:
    _ddpm_two_halves); forced on a small batch it must reproduce the single call bit for bit, guided or not."""
    gen = _gen(T=20)
    sizes = [9, 17, 29, 5, 30, 12, 8, 21, 16, 25] if connectivity == "template" else [9, 12, 7, 5, 10, 6, 8, 11]
    props = torch.linspace(-1.0, 1.0, 15).reshape(1, 15)
    for target, scale in ((None, 1.0), (props, 2.0)):
        gen.concurrent_atoms = None
        xa, pa, _, _ = gen.sample_batch(sizes, target, scale, 1.0, seed=5, connectivity=connectivity)
        gen.concurrent_atoms = 1
        xb, pb, _, _ = gen.sample_batch(sizes, target, scale, 1.0, seed=5, connectivity=connectivity)
        assert torch.equal(xa, xb) and torch.equal(pa, pb)
        assert torch.isfinite(xa).all() and torch.isfinite(pa).all()
    assert gen.model.engine().set_concurrent_calls(1) == 1    # the hint is restored after the call
    # every call reuses ONE pair of side streams (a fresh pair per call was seen to land on one hardware queue)
    assert len(gen._half_streams) == 2
    first = [s.cuda_stream for s in gen._half_streams]
    gen.sample_batch(sizes, None, 1.0, 1.0, seed=6, connectivity=connectivity)
    assert [s.cuda_stream for s in gen._half_streams] == first


def test_philox_normal_statistics_and_determinism():
    gen = _gen(T=20)
    eng = gen.model.engine()
    a = eng.philox_normal(1 << 20, 7, 3, 0)
    b = eng.philox_normal(1 << 20, 7, 3, 0)
    c = eng.philox_normal(1 << 19, 7, 3, 0, offset=1 << 19)
    assert torch.equal(a, b) and torch.equal(a[1 << 19:], c)
    assert abs(float(a.mean())) < 5e-3 and abs(float(a.var()) - 1.0) < 1e-2
    assert float(a.abs().max()) < 7.0 and torch.isfinite(a).all()
    d = eng.philox_normal(1 << 20, 7, 4, 0)
    assert abs(float((a * d).mean())) < 5e-3


def test_validity_gate_matches_oracle():
    gen = _gen(T=20)
    eng = gen.model.engine()
    g = torch.Generator().manual_seed(5)
    cases = []
    for k in range(12):
        n = [2, 5, 9, 16, 29, 64][k % 6]
        x = torch.randn(n, 119, generator=g)
        if k % 2 == 0:
            x[torch.arange(n), torch.randint(0, 119, (n,), generator=g)] += 12.0   # dominant type
        pos = torch.randn(n, 3, generator=g) * (3.0 if k % 3 else 0.05)
        if k == 7:
            x[0, 0] = float("nan")
        cases.append((x, pos))
    # a molecule that passes: grid points 1.5 apart, one dominant type per atom
    n = 8
    grid = torch.tensor([[i, j, k] for i in range(2) for j in range(2) for k in range(2)], dtype=torch.float32) * 1.5
    xg = torch.zeros(n, 119)
    xg[:, 6] = 10.0
    cases.append((xg, grid))
    x = torch.cat([c[0] for c in cases]).cuda()
    pos = torch.cat([c[1] for c in cases]).cuda()
    mol_ptr = torch.tensor(np.cumsum([0] + [c[0].shape[0] for c in cases]), dtype=torch.int32).cuda()
    acc = eng.validity_gate(x, pos, mol_ptr, len(cases)).cpu().tolist()
    want = [int(oracle.validity_gate(c[0], c[1], c[0].shape[0])) for c in cases]
    assert acc == want and want[-1] == 1


def test_generator_reference_api_runs_and_counts_attempts():
    gen = _gen(T=10)
    np.random.seed(0)
    mols = gen.generate_with_research_protocols(num_molecules=6, atom_range=(10, 40), use_ddim=False)
    assert gen.last_stats["attempted"] == 6 and gen.last_stats["accepted"] == len(mols)
    for m in mols:
        assert set(m) >= {"atom_features", "positions", "num_atoms", "generation_id", "generated_by_model"}
    r = gen.ddpm_sampling_fixed(12, None, 2.0, 1.0)
    assert r is None or (r[0].shape == (12, 119) and r[1].shape == (12, 3))
    r = gen.ddim_sampling_fixed(12, torch.rand(15), 5, 2.0, 1.0)
    assert r is None or r[0].shape == (12, 119)


@pytest.mark.parametrize("tag", ["ddim_plain", "ddim_cfg", "euler_plain", "euler_cfg"])
def test_alternate_samplers_match_reference_fixture(tag):
    """model1.py:890-966 (`sample`) and train.py:786-863 (trainer Euler sampler): both molecules of
    the fixture in ONE batch on the HIP path, fed the reference's own draws."""
    import copy
    from diffusion_model_ligandmol_generation_amd import alt_samplers
    f = golden("alt_samplers.npz")
    S, n, steps, _ = [int(v) for v in f[tag + "_meta"]]
    model = copy.deepcopy(seed0_model()).cuda().eval()
    props = torch.from_numpy(f["props"]) if tag.endswith("cfg") else None
    scale = float(f[tag + "_scale"])
    x0 = torch.from_numpy(f[tag + "_x0"]).reshape(S * n, 119)
    p0 = torch.from_numpy(f[tag + "_pos0"]).reshape(S * n, 3)
    nz = torch.from_numpy(f[tag + "_noise"])          # [S, steps, n, 3]
    noise = lambda i, shape: nz[:, i].reshape(S * n, 3)  # noqa: E731
    if tag.startswith("ddim"):
        pos, atom_pred = alt_samplers.ddim_positions(model, S, props, scale, steps, n, noise=noise, init=(x0, p0))
        torch.testing.assert_close(atom_pred.cpu().view(S, n, 119), torch.from_numpy(f[tag + "_atom_pred"]),
                                   rtol=1e-4, atol=1e-4)
        tol = 1e-4
    else:
        assert alt_samplers.chain_edges(n, ring=True) == f[tag + "_edges"].T.tolist()
        pos = alt_samplers.euler_positions(model, S, props, scale, steps, float(f[tag + "_temperature"]), n,
                                           noise=noise, init=(x0, p0))
        tol = 2e-3   # predictions are multiplied by sigma_t - sigma_next (up to ~30): rounding grows
    torch.testing.assert_close(pos.cpu().view(S, n, 3), torch.from_numpy(f[tag + "_pos"]), rtol=tol, atol=tol)


def test_alternate_sampler_wrappers_run():
    import copy
    from diffusion_model_ligandmol_generation_amd.train import ResearchAccurateTrainer
    model = copy.deepcopy(seed0_model()).cuda()
    mols = model.sample(num_samples=3, num_steps=5, max_atoms=12, seed=5)
    assert len(mols) == 3 and mols[0]["positions"].shape == (12, 3) and mols[0]["atom_types"].shape == (12,)
    again = model.sample(num_samples=3, num_steps=5, max_atoms=12, seed=5)
    assert np.array_equal(mols[1]["positions"], again[1]["positions"])
    tr = ResearchAccurateTrainer(model, [], [], [])
    out = tr.generate_research_validated_molecules(num_samples=2, num_steps=4, seed=1)
    assert len(out) == 2 and out[0]["positions"].shape == (15, 3) and np.isfinite(out[0]["positions"]).all()


@pytest.mark.parametrize("guided", [False, True])
def test_ddim_library_loop_equals_host_loop(guided):
    """ligdiff_ddim_sample (tables + graph replay) against the step-by-step host loop that the
    trajectory fixture pins: bitwise equal states and gate flags."""
    from diffusion_model_ligandmol_generation_amd.generation import ResearchValidatedGenerator
    m = copy.deepcopy(seed0_model()).cuda().eval()
    gen = ResearchValidatedGenerator(m)
    sizes = [12, 29, 7, 33]
    props = torch.rand(15) if guided else None
    with torch.no_grad():
        xa, pa, _, ra = gen.sample_batch(sizes, props, 2.0, 1.0, use_ddim=True, num_steps=9, seed=11)
        xb, pb, _, rb = gen.sample_batch(sizes, props, 2.0, 1.0, use_ddim=True, num_steps=9, seed=11,
                                         trace=lambda t, x, p: None)
        xc, pc, _, rc = gen.sample_batch(sizes, props, 2.0, 1.0, use_ddim=True, num_steps=9, seed=11, use_graph=False)
    assert torch.equal(xa, xb) and torch.equal(pa, pb) and torch.equal(ra, rb)
    assert torch.equal(xa, xc) and torch.equal(pa, pc) and torch.equal(ra, rc)
    assert torch.isfinite(xa).all()


def test_guided_one_pass_equals_two_pass_bitwise(monkeypatch):
    """Classifier-free guidance as ONE forward over [conditional; unconditional copy] against two
    separate forwards (LIGDIFF_CFG_TWO_PASS=1): molecules never interact, so every bit must agree -
    DDPM and DDIM library loops, ragged sizes, graph replay on and off."""
    gen = _gen(T=12)
    sizes = [9, 17, 29, 5, 33]
    props = torch.rand(len(sizes), 15, generator=torch.Generator().manual_seed(4))
    out = {}
    for two_pass in (False, True):
        if two_pass:
            monkeypatch.setenv("LIGDIFF_CFG_TWO_PASS", "1")
        else:
            monkeypatch.delenv("LIGDIFF_CFG_TWO_PASS", raising=False)
        a = gen.sample_batch(sizes, props, 2.5, 0.8, seed=21, use_graph=True)
        b = gen.sample_batch(sizes, props, 2.5, 0.8, seed=21, use_graph=False)
        c = gen.sample_batch(sizes, props, 2.5, 1.0, seed=21, use_ddim=True, num_steps=6)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        out[two_pass] = (a[0], a[1], c[0], c[1])
    for u, v in zip(out[False], out[True]):
        assert torch.isfinite(u).all() and torch.equal(u, v)


@pytest.mark.parametrize("split", [False, True])
def test_large_ragged_molecules_sample_consistently(split):
    """Molecules spanning one to four 32-atom attention tiles (up to the 128-atom limit): graph replay,
    host loop and a two-shard split agree bitwise, in either GEMM mode."""
    from diffusion_model_ligandmol_generation_amd import _lib
    lib = _lib.load()
    prev = lib.ligdiff_set_split_gemm(1 if split else 0)
    try:
        gen = _gen(T=6, max_atoms=128)
        sizes = [70, 128, 33, 96, 12]
        xa, pa, _, _ = gen.sample_batch(sizes, None, 1.0, 1.0, seed=5, use_graph=True)
        xb, pb, _, _ = gen.sample_batch(sizes, None, 1.0, 1.0, seed=5, use_graph=False)
        x1, p1, _, _ = gen.sample_batch(sizes[:2], None, 1.0, 1.0, seed=5, atom_offset=0)
        x2, p2, _, _ = gen.sample_batch(sizes[2:], None, 1.0, 1.0, seed=5, atom_offset=sum(sizes[:2]))
    finally:
        lib.ligdiff_set_split_gemm(prev)
    assert torch.isfinite(xa).all() and torch.isfinite(pa).all()
    assert torch.equal(xa, xb) and torch.equal(pa, pb)
    assert torch.equal(torch.cat([x1, x2]), xa) and torch.equal(torch.cat([p1, p2]), pa)


@pytest.mark.parametrize("tag", ["dict_guided", "dict_plain"])
def test_dict_valued_target_properties_match_reference(tag):
    """Dict-valued target_properties (generation.py:689-757): the reference's model call fails on them and returns
    zero predictions (model1.py:337, :757-764); the generator reproduces exactly that (zero conditional prediction,
    guidance combination as generation.py:442-443).  Final state of a 20-step DDPM against the reference's
    (tests/golden/quirk.npz), with the reference's own draws."""
    f = golden("quirk.npz")
    T, n, seed = [int(v) for v in f[tag + "_meta"]]
    gen = _gen(T)
    noise = SeededNoise(seed, n, T)
    tp = {"molecular_weight": 350.0, "logp": 2.5, "qed": 0.8}
    x, pos, _, _ = gen.sample_batch([n], tp, float(f[tag + "_scale"]), 0.9, noise=noise)
    fx, fp = torch.from_numpy(f[tag + "_final_x"]), torch.from_numpy(f[tag + "_final_pos"])
    torch.testing.assert_close(x.cpu(), fx, rtol=1e-3, atol=1e-3 * float(fx.abs().max()))
    torch.testing.assert_close(pos.cpu(), fp, rtol=1e-3, atol=1e-3 * float(fp.abs().max()))
    # and through the public entry point with the library's own noise: runs, returns the reference's keys
    out = gen.generate_with_research_protocols(num_molecules=3, target_properties=tp, atom_range=(10, 16), use_ddim=False)
    assert gen.last_stats["attempted"] == 3 and all(o["target_properties"] == tp for o in out)
