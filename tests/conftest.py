import os
import sys

import numpy as np
import pytest
import torch

# The library picks the fused chain kernels of the inference path by how well a launch fills the chip
# (small batches run the separate GEMM launches).  The GPU parity tests are small on purpose, so they ask
# for the fused kernels at every size; tests/test_gpu_chain.py compares the two paths directly.
os.environ.setdefault("LIGDIFF_CHAIN", "force")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


_MODELS = {}


def seed0_model(**kw):
    """The build's model constructed under torch.manual_seed(0); weights are verified
    against the checksums the reference's own seed-0 model produced."""
    key = tuple(sorted(kw.items()))
    if key not in _MODELS:
        from diffusion_model_ligandmol_generation_amd.model import ResearchAccurateDiffusionModel
        torch.manual_seed(0)
        m = ResearchAccurateDiffusionModel(**kw).eval()
        ck = golden("weights_seed0_checksums.npz")
        sd = m.state_dict()
        for k in ck.files:
            if k.startswith("moldiff_consistency.") and ("max_valences" in k or "covalent_radii" in k):
                continue  # RDKit-derived constants: inputs of a checkpoint, not of the seed
            if k in ("betas", "alphas", "alphas_cumprod", "sqrt_alphas_cumprod",
                     "sqrt_one_minus_alphas_cumprod"):
                continue  # schedule buffers: not seeded weights (checked against schedule.npz)
            v = sd[k].double()
            got = np.array([float(v.sum()), float(v.abs().sum()), float(v.flatten()[0]), float(v.flatten()[-1])])
            # first / last element exact; the fp64 sums only to reduction-order rounding
            assert np.array_equal(got[2:], ck[k][2:]) and np.allclose(got[:2], ck[k][:2], rtol=1e-11, atol=1e-13), \
                f"seed-0 weight {k} differs from the reference's"
        _MODELS[key] = m
    return _MODELS[key]


def seed0_state_dict(**kw):
    return {k: v.detach().clone() for k, v in seed0_model(**kw).state_dict().items()}


@pytest.fixture(scope="session")
def lib():
    from diffusion_model_ligandmol_generation_amd import _lib
    return _lib.load()


def no_dropout(model):
    """Sets every dropout probability of the attention layers to 0, the way the train-mode fixtures
    were generated from the reference (tests/golden/make_golden.py: attention.dropout and the two
    nn.Dropout of ff_network)."""
    for lyr in model.dit_layers:
        lyr.attention.dropout = 0.0
        lyr.ff_network[2].p = 0.0
        lyr.ff_network[4].p = 0.0
    model.moldiff_consistency.atom_predictor[5].p = 0.0       # tests/golden/make_golden_alt.py does the same
    model.moldiff_consistency.bond_predictor[5].p = 0.0
    return model
