#!/usr/bin/env python3
"""Headline benchmark: sampled molecules/s for the 1000-step DDPM reverse loop.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" is one full pass of the hot path over one batch: 1024 synthetic 29-atom ligands
with the generator's bond template, initial noise, 1000 reverse steps (score-network
forward + fused posterior update, hipGraph-replayed) and the validity gates, all on
device, fp32.  With N > 1 (launched by torch.distributed.run, one rank per GPU) every rank
samples its own 1024 molecules - molecules are independent, so there is no data-path
collective and scaling is weak; the Philox noise is keyed by the global atom index.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MOLS_PER_GPU = 1024
ATOMS = 29
T_STEPS = 1000
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X dense fp32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X dense bf16 matrix peak (2.5 PFLOP/s)
PEAK_HBM_GBS = 8000.0            # MI355X HBM3E peak
# HBM-side bytes per launch at this workload from rocprofv3 PMC passes (2 x FETCH_SIZE + WRITE_SIZE,
# the gfx950 correction for wide reads); the counters cannot be read from inside this process, so
# these are the figures of the committed profile (profiles/r1_b_pmc_summary.csv), not live ones.
FFN_UP_TRAFFIC_BYTES = 2 * 43616.9e3 * 1.024 + 118784.0e3 * 1.024
FFN_UP_SPLIT_TRAFFIC_BYTES = 2 * 49581.6e3 * 1.024 + 118784.0e3 * 1.024   # profiles/r1_e_split_pmc_summary.csv
AGGREGATE_TRAFFIC_BYTES = 2 * 35435.5e3 * 1.024 + 30404.5e3 * 1.024


def build_model(device):
    from diffusion_model_ligandmol_generation_amd import ResearchAccurateDiffusionModel
    torch.manual_seed(0)
    return ResearchAccurateDiffusionModel().to(device).eval()


def cpu_baseline(n_steps_cap=None):
    """The oracle (CPU restatement, validated against the reference's outputs) running the
    loop the way the reference does: one molecule per forward, padded to max_atoms=100."""
    import oracle
    from diffusion_model_ligandmol_generation_amd import ResearchAccurateDiffusionModel
    torch.manual_seed(0)
    m = ResearchAccurateDiffusionModel().eval()
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    sched = oracle.cosine_schedule(T_STEPS)
    g = torch.Generator().manual_seed(1234)
    noise = lambda shape: torch.randn(*shape, generator=g)  # noqa: E731
    cores = torch.get_num_threads()
    n_mols = 3
    t0 = time.time()
    with torch.no_grad():
        for _ in range(n_mols):
            oracle.ddpm_sample(sd, sched, ATOMS, noise, None, 1.0, 1.0, max_atoms=100, pad=True)
    dt = time.time() - t0
    return {"value": n_mols / dt, "unit": "molecules/s", "cores": cores, "kind": "port",
            "sample": f"{n_mols} molecules x {ATOMS} atoms x {T_STEPS} DDPM steps, one molecule per forward padded to "
                      f"max_atoms=100 as the reference runs it ({dt:.1f} s)"}


def cpu_baseline_batched(n_mols=64, n_steps=150):
    """Fairness variant of the CPU baseline: the same restatement on a packed batch of 64 molecules
    (no padding, one forward per step for the whole batch), timed over a bounded number of reverse
    steps and scaled to the 1000-step loop."""
    import numpy as np
    import oracle
    from diffusion_model_ligandmol_generation_amd import ResearchAccurateDiffusionModel
    torch.manual_seed(0)
    m = ResearchAccurateDiffusionModel().eval()
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    sched = oracle.cosine_schedule(T_STEPS)
    g = torch.Generator().manual_seed(1234)
    ei1 = np.asarray(oracle.template_connectivity(ATOMS)[0], dtype=np.int64).reshape(-1, 2).T
    mols = [{"x": torch.randn(ATOMS, 119, generator=g).numpy(), "pos": torch.randn(ATOMS, 3, generator=g).numpy(),
             "edge_index": ei1} for _ in range(n_mols)]
    c = oracle.collate(mols)
    x, pos, ei, b = (torch.from_numpy(c[k]) for k in ("x", "pos", "edge_index", "batch"))
    t0 = time.time()
    with torch.no_grad():
        for t in range(T_STEPS - 1, T_STEPS - 1 - n_steps, -1):
            ex, ep, _ = oracle.score_network_forward(sd, x, pos, ei, b, torch.full((n_mols,), float(t)), None,
                                                     pad_to_max_atoms=False)
            x = oracle.ddpm_step(x, ex, t, sched, torch.randn(x.shape, generator=g), 1.0)
            pos = oracle.ddpm_step(pos, ep, t, sched, torch.randn(pos.shape, generator=g), 1.0)
    dt = time.time() - t0
    return {"value": n_mols / (dt * T_STEPS / n_steps), "unit": "molecules/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{n_mols} molecules x {ATOMS} atoms packed in one batch, {n_steps} of {T_STEPS} DDPM steps "
                      f"({dt:.1f} s), scaled to the full loop"}


def mode_name(split):
    return ("fp32 operands as 3 bf16 terms each, 6 products on the bf16 matrix pipe, fp32 accumulate (opt-in)"
            if split else "v_mfma_f32_32x32x2_f32 (fp32 matrix pipe)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # LIGDIFF_BENCH_REHEARSE=1: run the N > 1 code path on a box with ONE GPU (all ranks on cuda:0, gloo
    # for the barrier and the max-reduce); the numbers of such a run mean nothing.
    rehearse = os.environ.get("LIGDIFF_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from diffusion_model_ligandmol_generation_amd.generation import ResearchValidatedGenerator
    from diffusion_model_ligandmol_generation_amd.engine import template_graph
    model = build_model(dev)
    gen = ResearchValidatedGenerator(model)
    eng = model.engine()
    sizes = [ATOMS] * MOLS_PER_GPU
    atoms_per_rank = ATOMS * MOLS_PER_GPU

    def one_pass(i):
        # global molecule range of this rank in pass i -> Philox offsets independent of N
        offset = (i * world + rank) * atoms_per_rank
        with torch.no_grad():
            x, pos, graph, rejected = gen.sample_batch(sizes, None, 1.0, 1.0, seed=2024, atom_offset=offset)
            accept = eng.validity_gate(x, pos, graph.mol_ptr, graph.n_mols)
        return x, pos, graph, accept

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_run(warmup, steps):
        """W untimed + K timed passes between barriers; returns (max-over-ranks seconds, accepted, last x, pos)."""
        for i in range(warmup):
            one_pass(1000 + i)
        fence()
        t0 = time.time()
        accepted = 0
        x = pos = None
        for i in range(steps):
            x, pos, graph, accept = one_pass(i)
            accepted += int(accept.sum().item())
        fence()
        dt = time.time() - t0
        if world > 1:
            tt = torch.tensor([dt], device="cpu" if rehearse else dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, accepted, x, pos

    # GEMM mode of the library: fp32 matrix pipe unless LIGDIFF_GEMM_SPLIT=1 is in the environment.
    # The headline run uses that mode; the other mode is timed afterwards and reported beside it.
    split_mode = eng.set_split_gemm(False)
    eng.set_split_gemm(split_mode)
    dt, accepted, x, pos = timed_run(args.warmup, args.steps)
    finite = bool(torch.isfinite(x).all() and torch.isfinite(pos).all())
    # (every rank runs it, so the barriers inside stay matched; a failure here must not lose the headline)
    alt_error = None
    try:
        eng.set_split_gemm(not split_mode)
        dt_alt, _, x_alt, pos_alt = timed_run(1, args.steps)
    except Exception as e:   # noqa: BLE001
        alt_error = repr(e)
    finally:
        eng.set_split_gemm(split_mode)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * MOLS_PER_GPU * args.steps / dt
        # ---- roofline of the dominant kernel (fp32 MFMA GEMM: FFN up-projection of a DiT layer)
        N = atoms_per_rank
        t_index = torch.full((1,), 500, dtype=torch.int32, device=dev)
        graph, _, _ = template_graph(sizes, dev)
        def ffn_up_roofline(split):
            eng.set_split_gemm(split)
            eng.time_launch("dit1.ffn_up", graph, x, pos, t_index, reps=2)           # warm
            ms = eng.time_launch("dit1.ffn_up", graph, x, pos, t_index, reps=10)
            eng.set_split_gemm(split_mode)
            flops = 2.0 * N * 256 * 1024
            if not split:
                ach = flops / (ms * 1e-3) / 1e12
                return {"kernel": "gemm_cs_kernel<APtrPlain, EpiBiasAct<8,GELU>> (dit*.ffn_up, M=29696 K=256 N=1024)",
                        "bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": FFN_UP_TRAFFIC_BYTES, "avg_launch_ms": ms}
            # six bf16 x bf16 products per fp32 product: the matrix pipe executes 6x the algorithmic flops
            ach = 6.0 * flops / (ms * 1e-3) / 1e12
            return {"kernel": "gemm_cs_split_kernel<APtrPlain, EpiBiasAct<8,GELU>> (dit*.ffn_up, M=29696 K=256 N=1024)",
                    "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / PEAK_BF16_MFMA_TFLOPS, "traffic": FFN_UP_SPLIT_TRAFFIC_BYTES, "avg_launch_ms": ms,
                    "fp32_equivalent_tflops": flops / (ms * 1e-3) / 1e12,
                    "note": "achieved counts the six bf16 MFMA products issued per fp32 product"}
        roofline = ffn_up_roofline(split_mode)
        if alt_error is None:
            alt = {"gemm_mode": mode_name(not split_mode),
                   "value": world * MOLS_PER_GPU * args.steps / dt_alt, "unit": "molecules/s",
                   "ms_per_step": dt_alt / args.steps * 1e3,
                   "max_abs_diff_vs_headline": {"x": float((x_alt - x).abs().max()),
                                                "pos": float((pos_alt - pos).abs().max())},
                   "roofline": ffn_up_roofline(not split_mode)}
        else:
            alt = {"gemm_mode": mode_name(not split_mode), "error": alt_error}
        # ---- the aggregation kernel against the HBM roofline (north-star target kernel)
        E = graph.n_edges
        ms_agg = eng.time_launch("e3_1.aggregate", graph, x, pos, t_index, reps=10)
        bytes_agg = E * 1024.0 + N * 1024.0 + E * 16.0 * 2 + N * 36.0          # msg rows in, agg rows out, wrel, pos
        agg_gbs = bytes_agg / (ms_agg * 1e-3) / 1e9
        roofline_agg = {"kernel": "k_aggregate (e3_*.aggregate)", "bound": "hbm", "achieved": agg_gbs,
                        "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": agg_gbs / PEAK_HBM_GBS, "traffic": AGGREGATE_TRAFFIC_BYTES,
                        "avg_launch_ms": ms_agg}
        out = {
            "metric": "sampled molecules/sec (1000-step DDPM)", "value": value, "unit": "molecules/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (operands split into 3 bf16 terms, 6 bf16 MFMA products, fp32 accumulate)" if split_mode else "f32",
            "data": "synthetic (seeded template ligands, random-init weights under torch.manual_seed(0))",
            "config": {"workload": "configs[1]: 29-atom template ligands, batch 1024 per GPU, 1000-step DDPM, fp32",
                       "molecules_per_gpu": MOLS_PER_GPU, "atoms_per_molecule": ATOMS,
                       "edges_per_molecule": 62, "ddpm_steps": T_STEPS, "guidance": "none",
                       "gemm_mode": mode_name(split_mode),
                       "parallelism": f"molecule-sharded x{world}, no collectives"},
            "attempted_molecules": world * MOLS_PER_GPU * args.steps,
            "accepted_rank0": accepted, "finite": finite,
            "roofline": roofline, "roofline_aggregation": roofline_agg,
            # the same K passes in the library's other GEMM mode (not the headline; see config.gemm_mode)
            "alt_gemm_mode": alt,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            out["cpu_baseline_batched"] = cpu_baseline_batched()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
