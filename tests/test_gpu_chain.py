"""The fused chain kernels of the inference path (ffn / edge / node / attention-layer tail, csrc/*_chain.h)
against the separate-launch kernels they replace: the same forwards and one sampling call in two processes,
one with LIGDIFF_CHAIN=force, one with LIGDIFF_CHAIN=0 (and a third in the default by-fill mode, which must pick the
separate launches at these sizes bit for bit).  Both paths are pinned to the reference by the other GPU tests (each runs with
the default, fused, path); this one compares them directly, at shapes that exercise partial workgroups,
one-atom-short tiles, per-molecule timesteps, property conditioning and fully connected edges."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, name, extra_env):
    out = str(tmp_path / f"{name}.npz")
    env = dict(os.environ)
    env.pop("LIGDIFF_NO_CHAIN", None)
    env.pop("LIGDIFF_CHAIN", None)
    env.update(extra_env)
    r = subprocess.run([sys.executable, os.path.join(HERE, "_chain_ab.py"), out], env=env, capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


def test_chain_kernels_match_separate_launches(tmp_path):
    fused = _run(tmp_path, "fused", {"LIGDIFF_CHAIN": "force"})
    plain = _run(tmp_path, "plain", {"LIGDIFF_CHAIN": "0"})
    rows = _run(tmp_path, "rows", {"LIGDIFF_CHAIN": "force", "LIGDIFF_CHAIN_MOL": "0"})   # attention as its own launch
    auto = _run(tmp_path, "auto", {})   # by fill: these batches are far too small for the fused kernels
    for k in plain.files:
        assert np.array_equal(auto[k], plain[k]), k
    assert set(fused.files) == set(plain.files)
    # case 0 (40 x 29 atoms) takes the molecule-per-wave form of the attention-layer kernel when it is allowed
    assert not np.array_equal(fused["c0_atom"], rows["c0_atom"])
    for k in fused.files:
        if not k.startswith("c0_") and not k.startswith("s_"):
            assert np.array_equal(fused[k], rows[k]), k
    for name, other in (("plain", plain), ("rows", rows)):
        for k in fused.files:
            a, b = fused[k], other[k]
            assert a.shape == b.shape and np.isfinite(a).all() and np.isfinite(b).all(), (name, k)
            # fp32 summation orders differ between the paths: tolerance relative to the tensor's scale
            # (the sampled state after 25 reverse steps accumulates 25 forwards' worth of it)
            scale = max(1.0, float(np.abs(b).max()))
            tol = (2e-4 if k.startswith("s_") else 2e-5) * scale
            assert float(np.abs(a - b).max()) <= tol, (name, k, float(np.abs(a - b).max()), scale)


def test_chain_mode_switch_in_process():
    """ligdiff_set_chain_mode: the run-time override of LIGDIFF_CHAIN returns the previous setting, rejects
    modes out of range, and the two forms it selects agree on one sampling call."""
    import torch
    from diffusion_model_ligandmol_generation_amd import ResearchAccurateDiffusionModel
    from diffusion_model_ligandmol_generation_amd.generation import ResearchValidatedGenerator
    torch.manual_seed(0)
    model = ResearchAccurateDiffusionModel().cuda().eval()
    eng = model.engine()
    gen = ResearchValidatedGenerator(model)
    prev = eng.set_chain_mode(0)
    try:
        assert prev == -1
        with pytest.raises(ValueError):
            eng.set_chain_mode(3)
        assert eng.set_chain_mode(0) == 0   # a rejected mode leaves the setting alone
        x0, p0, _, _ = gen.sample_batch([29] * 12, None, 1.0, 1.0, num_steps=10, seed=3)
        assert eng.set_chain_mode(2) == 0
        x2, p2, _, _ = gen.sample_batch([29] * 12, None, 1.0, 1.0, num_steps=10, seed=3)
    finally:
        eng.set_chain_mode(prev)
    assert torch.isfinite(x2).all() and torch.isfinite(p2).all()
    assert not torch.equal(x0, x2)          # different kernels, different summation orders
    for a, b in ((x0, x2), (p0, p2)):
        scale = max(1.0, float(b.abs().max()))
        assert float((a - b).abs().max()) <= 1e-4 * scale


def test_half_tile_gemms_are_bitwise_the_full_tile_ones(tmp_path):
    """Column-split GEMM launches of few rows run as 64-row workgroups (csrc/launch.h: half_tiles); a row's products
    and sums are the same instructions in the same order as in the 128-row form, so every output of the separate-launch
    path (forwards with ragged rows, properties, fully connected edges, and a sampling call) must agree bit for bit with
    a process in which LIGDIFF_GEMM_TILE64=0 keeps the 128-row workgroups."""
    half = _run(tmp_path, "half", {"LIGDIFF_CHAIN": "0"})
    full = _run(tmp_path, "full", {"LIGDIFF_CHAIN": "0", "LIGDIFF_GEMM_TILE64": "0"})
    assert set(half.files) == set(full.files)
    for k in half.files:
        assert np.isfinite(half[k]).all(), k
        assert np.array_equal(half[k], full[k]), k
