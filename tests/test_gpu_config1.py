"""GPU: the end-to-end flow of the reference's driver at batch 64 (main.py:187-274: split, loaders, model, one
training step, 100-step DDPM sampling) through the C-ABI, with the CPU oracle run on the same draws beside it
(tools/run_config1.py, the harness bench.py reports as ``config1``).  The harness only prints its differences;
this test fails when they regress.

Parity bar: total loss rtol 1e-4, every clipped parameter gradient within 1e-4 of the tensor's largest entry,
sampled trajectory (100 reverse steps, the library's own Philox draws fed to the oracle) within 1e-3 of the
state's scale."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_config1_flow_matches_oracle():
    import run_config1
    out = run_config1.run()
    ts, sm = out["train_step"], out["sampling"]
    assert ts["molecules"] == 64
    assert ts["rel_loss_diff"] < 1e-4, ts
    assert ts["worst_rel_gradient_diff"] < 1e-4, ts
    assert sm["finite"] and sm["molecules"] == 64
    assert sm["oracle_cpu"]["max_rel_trajectory_drift_vs_hip"] < 1e-3, sm
