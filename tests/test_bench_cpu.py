"""Host logic of bench.py that needs no GPU: the launcher's refusal when fewer GPUs are visible than
--gpus asks for, the traffic table read from the committed rocprofv3 PMC summaries, and the seeded sizes
of the ragged legs."""
import importlib.util
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("ligdiff_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_traffic_table_comes_from_the_committed_pmc_summaries():
    bench = _bench()
    table, source = bench.profile_traffic()
    assert source is not None and "profiles/" in source and "split" not in source
    # the kernels bench.py prices: fused attention-layer kernel and the aggregation kernel
    chain = bench.traffic_of(table, "dit_chain_kernel<0, true")
    agg = bench.traffic_of(table, "k_aggregate")
    assert chain is not None and 100e6 < chain < 500e6          # 152 MB algorithmic per launch
    assert agg is not None and 96.4e6 <= agg < 130e6              # 96.4 MB algorithmic per launch
    assert bench.traffic_of(table, "no_such_kernel") is None


def test_ragged_leg_sizes_are_seeded_and_in_range():
    bench = _bench()
    a, b = bench.uniform_sizes(512, 10, 64, 1234), bench.uniform_sizes(512, 10, 64, 1234)
    assert a == b and min(a) >= 10 and max(a) <= 64 and len(a) == 512
    assert bench.uniform_sizes(512, 10, 64, 1235) != a


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason="needs a box with fewer than two GPUs")
def test_launcher_refuses_more_ranks_than_gpus():
    """`bench.py --gpus 2` outside torch.distributed.run starts the ranks itself; with fewer than two GPUs
    visible it must exit non-zero before anything touches a GPU, and print no JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LIGDIFF_BENCH_REHEARSE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "--gpus 2" in r.stderr and r.stdout.strip() == ""
