#!/usr/bin/env python3
"""Headline benchmark: sampled molecules/s for the 1000-step DDPM reverse loop.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" is one full pass of the hot path over one batch: 1024 synthetic 29-atom ligands
with the generator's bond template, initial noise, 1000 reverse steps (score-network
forward + fused posterior update, hipGraph-replayed) and the validity gates, all on
device, fp32.  With N > 1 every rank (one process per GPU) samples its own 1024 molecules -
molecules are independent, so there is no data-path collective and scaling is weak; the
Philox noise is keyed by the global atom index.  Rank 0 prints ONE JSON line.

Launching: under ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`` each
process is one rank (RANK / LOCAL_RANK / WORLD_SIZE from the environment).  Started directly as
``python bench.py --gpus N`` with N > 1 and no WORLD_SIZE, this process touches no GPU: it starts
the N ranks itself through torch.distributed.run and exits with their status.

Beside the headline the same line carries three bounded legs on the other named workloads
(``config3_train_step``, ``config4_fc_sampling``, ``config5_ragged_split_sweep``); they run after
the headline has been measured and can neither change nor (through a watchdog) lose it.
"""
import argparse
import csv
import glob
import json
import os
import socket
import subprocess
import sys
import threading
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
EXTRAS_BUDGET_S = 240.0          # the extra legs are abandoned (headline still printed) beyond this


# ------------------------------------------------------------------ launcher
def spawn_ranks(args, passthrough):
    """--gpus N > 1 without a torch.distributed environment: start N ranks (one per GPU) before
    anything here touches a GPU, pass their output through and return their exit status."""
    rehearse = os.environ.get("LIGDIFF_BENCH_REHEARSE") == "1"
    have = torch.cuda.device_count()          # counts devices without initialising the runtime
    if not rehearse and have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + passthrough
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------ counters from the committed profile
def profile_traffic(shape="headline"):
    """HBM-side bytes per launch from the newest committed rocprofv3 PMC summary taken at ``shape``
    (profiles/r<round>_<letter>[_<shape>]_pmc_summary.csv: per-kernel means of FETCH_SIZE and WRITE_SIZE in KB,
    taken in separate passes; no shape tag = the headline shape, 1024 x 29-atom template ligands; "train" the training
    step at that shape; "fc", "train3", "ragged5" the shapes of the extra legs, tools/pmc_shape.py).  On gfx950
    FETCH_SIZE counts half the bytes of a wide streaming read, so read bytes = 2 x FETCH_SIZE; WRITE_SIZE is exact.
    The counters cannot be read from inside this process; a kernel that the summary does not list gets None.
    Returns (table, file name or None)."""
    import re
    best = None
    for path in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.csv")):
        m = re.fullmatch(r"r(\d+)_([a-z])(?:_(\w+?))?_pmc_summary\.csv", os.path.basename(path))
        if m is None or (m.group(3) or "headline") != shape:
            continue
        key = (int(m.group(1)), m.group(2))
        if best is None or key > best[0]:
            best = (key, path)
    if best is None:
        return {}, None
    vals, out = {}, {}
    with open(best[1]) as fh:
        for row in csv.DictReader(fh):
            vals.setdefault(row["Kernel_Name"], {})[row["Counter_Name"]] = float(row["Mean_Counter_Value_KB"])
    for name, c in vals.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            out[name] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    return out, os.path.relpath(best[1], ROOT)


def traffic_of(table, *needles):
    for name, v in table.items():
        if all(n in name for n in needles):
            return v
    return None


# ------------------------------------------------------------------ models and CPU baselines
def build_model(device, max_atoms=100, train=False):
    from diffusion_model_ligandmol_generation_amd import ResearchAccurateDiffusionModel
    torch.manual_seed(0)
    m = ResearchAccurateDiffusionModel(max_atoms=max_atoms).to(device)
    return m.train() if train else m.eval()


def cpu_baseline():
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


# ------------------------------------------------------------------ one rank
class Ctx:
    """Rank, device and the fenced timing helper shared by every leg."""

    def __init__(self, args):
        self.t_start = time.time()
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # LIGDIFF_BENCH_REHEARSE=1: run the N > 1 code path on a box with ONE GPU (all ranks on cuda:0, gloo
        # for the collectives); the numbers of such a run mean nothing.
        self.rehearse = os.environ.get("LIGDIFF_BENCH_REHEARSE") == "1"
        if self.rehearse:
            self.local_rank = 0
        self.dist = None
        if self.world > 1:
            import datetime
            import torch.distributed as dist
            self.dist = dist
            if self.rehearse:
                dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=600))
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank),
                                        timeout=datetime.timedelta(seconds=600))
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        self.group = self.dist.group.WORLD if self.dist is not None else None

    def fence(self):
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if self.dist is None:
            return seconds
        tt = torch.tensor([seconds], device="cpu" if self.rehearse else self.dev, dtype=torch.float64)
        self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
        return float(tt.item())

    def sum_over_ranks(self, value):
        if self.dist is None:
            return value
        tt = torch.tensor([float(value)], device="cpu" if self.rehearse else self.dev, dtype=torch.float64)
        self.dist.all_reduce(tt, op=self.dist.ReduceOp.SUM)
        return float(tt.item())

    def timed(self, fn, warmup, steps):
        """W untimed + K timed calls of fn(i) between barrier + synchronize fences; max over ranks."""
        for i in range(warmup):
            fn(10_000 + i)
        self.fence()
        t0 = time.time()
        last = None
        for i in range(steps):
            last = fn(i)
        self.fence()
        return self.max_over_ranks(time.time() - t0), last


def log(ctx, msg):
    """Progress line on stderr (stdout carries only the JSON line)."""
    print(f"[bench rank {ctx.rank}/{ctx.world} +{time.time() - ctx.t_start:6.1f}s] {msg}", file=sys.stderr, flush=True)


def uniform_sizes(n, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi + 1, (n,), generator=g).tolist()


# ------------------------------------------------------------------ extra legs
def leg_config3_train_step(ctx, steps=10, warmup=3, mols_per_rank=512):
    """BASELINE configs[2]: data-parallel training step on drug-like ragged ligands (10-64 atoms, bond
    templates), 512 molecules per GPU (global batch 4096 on 8 GPUs): EDM noise/loss step, forward
    with saved activations, backward, ONE flat gradient all-reduce over all ranks, clip, AdamW."""
    from diffusion_model_ligandmol_generation_amd.engine import template_graph
    from diffusion_model_ligandmol_generation_amd.train import ResearchAccurateTrainer
    dev = ctx.dev
    sizes = uniform_sizes(mols_per_rank, 10, 64, 1234 + ctx.rank)
    graph, ei, _ = template_graph(sizes, dev)
    N = graph.n_atoms
    g = torch.Generator().manual_seed(99 + ctx.rank)
    types = torch.tensor([1, 6, 7, 8, 9])[torch.multinomial(torch.tensor([.45, .35, .08, .10, .02]), N, True, generator=g)]

    class Batch:
        pass

    b = Batch()
    b.x = torch.nn.functional.one_hot(types, 119).float().to(dev)
    b.pos = (torch.randn(N, 3, generator=g) * 1.5).to(dev)
    b.edge_index, b.batch = ei, graph.batch
    model = build_model(dev, 100, train=True)
    trainer = ResearchAccurateTrainer(model, [], [], [], process_group=ctx.group)
    dt, _ = ctx.timed(lambda i: trainer.train_step(b), warmup, steps)
    # the same step with the forward and input-gradient GEMMs in split-operand mode (per call, through the model's
    # library options; the weight-gradient GEMMs stay on the fp32 pipe)
    model.library_options = model.engine().options(gemm_mode=1)
    try:
        dt_split, _ = ctx.timed(lambda i: trainer.train_step(b), warmup, steps)
    finally:
        model.library_options = None
    n_atoms_global = ctx.sum_over_ranks(N)
    out = {"workload": "configs[2]: ragged 10-64-atom template ligands, 512 molecules per GPU, data-parallel training "
                       "step (EDM loss, forward, backward, flat gradient all-reduce, clip, AdamW), fp32, train mode",
           "molecules_per_gpu": mols_per_rank, "global_batch": mols_per_rank * ctx.world,
           "atoms_global": int(n_atoms_global), "steps": steps, "warmup": warmup,
           "ms_per_step": dt / steps * 1e3, "molecules_per_s": mols_per_rank * ctx.world * steps / dt,
           "ms_per_step_split_operand_mode": dt_split / steps * 1e3,
           "collective": (("gloo (rehearsal)" if ctx.rehearse else "RCCL all-reduce, 26.4 MB fp32 per step")
                          if ctx.world > 1 else "none (one rank)")}
    if ctx.world > 1:      # the gradient all-reduce on its own: one flat fp32 buffer of the gradients' size
        from diffusion_model_ligandmol_generation_amd.parallel import allreduce_sum_
        numel = sum(p.grad.numel() for p in model.parameters() if p.grad is not None)
        flat = torch.zeros(numel, device=dev)
        dta, _ = ctx.timed(lambda i: allreduce_sum_(flat, ctx.group), 2, 5)
        out["allreduce"] = {"bytes": numel * 4, "ms": dta / 5 * 1e3,
                            "bus_GBps": 2.0 * (ctx.world - 1) / ctx.world * numel * 4 / (dta / 5) / 1e9}
    if ctx.rank == 0:      # dominant backward kernel: the FFN up-projection weight gradient
        eng = model.engine()
        u = torch.rand(graph.n_mols, device=dev)
        ms, (nout, kin, rows) = eng.time_backward_launch(graph, b.x, b.pos, u, wide_dw_index=1, reps=5)
        flops = 2.0 * nout * kin * rows
        ach = flops / (ms * 1e-3) / 1e12
        tr3, tr3_src = profile_traffic("train3")
        out["roofline"] = {"kernel": f"gemm_dw_kernel<4> (dW[{nout},{kin}] += dY^T X over {rows} rows: dit2.ffn_up weight gradient)",
                           "bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": traffic_of(tr3, "gemm_dw_kernel<4>"),
                           "traffic_source": tr3_src,
                           "timing_note": "the probed launch alone on the stream (ligdiff_backward_timed); inside a "
                                          "training step it runs on the side stream under the input-gradient chain",
                           "traffic_note": "mean over the 12 launches of this kernel in a step (different shapes), "
                                           "rocprofv3 counters at this leg's shape",
                           "avg_launch_ms": ms}
    return out


def fc_partial_rows(sizes):
    """Rows of per-slab partial sums the fused edge kernel writes for fully connected molecules: an atom's n - 1
    incoming edges are consecutive, slabs are 32 edges counted from the molecule's first edge."""
    rows = 0
    for n in sizes:
        d = n - 1
        for a in range(n):
            rows += (a * d + d - 1) // 32 - (a * d) // 32 + 1
    return rows


def leg_config4_fc_sampling(ctx, mols_per_rank=48):
    """BASELINE configs[3]: large ligands (65-128 atoms), every ordered atom pair an edge, 1000-step DDPM;
    the global molecule list is cut into contiguous ranges balanced by the sum of n^2."""
    from diffusion_model_ligandmol_generation_amd.generation import ResearchValidatedGenerator
    from diffusion_model_ligandmol_generation_amd.engine import full_graph
    from diffusion_model_ligandmol_generation_amd.parallel import shard_ranges
    dev = ctx.dev
    sizes = uniform_sizes(mols_per_rank * ctx.world, 65, 128, 4321)
    lo, hi = shard_ranges(sizes, ctx.world, cost="pairs")[ctx.rank]
    mine = sizes[lo:hi]
    model = build_model(dev, 128)
    gen = ResearchValidatedGenerator(model)
    eng = model.engine()
    offset = sum(sizes[:lo])
    # kernel forms fixed ONCE on the whole request, so that every rank's shard takes the forms (and produces the
    # bits) of a single call over all molecules
    plan = eng.plan(sum(sizes), sum(n * (n - 1) for n in sizes), len(sizes), max(sizes))

    def one_pass(i):
        with torch.no_grad():
            x, pos, graph, _ = gen.sample_batch(mine, None, 1.0, 1.0, seed=77 + i, atom_offset=offset, connectivity="full",
                                                options=plan)
            acc = eng.validity_gate(x, pos, graph.mol_ptr, graph.n_mols)
        return x, pos, graph, acc

    dt, (x, pos, graph, acc) = ctx.timed(one_pass, 0, 1)
    # the same loop over its last 20 timesteps only (t = 19 .. 0): the state is still finite at the end of it, so
    # this figure does not sit on Inf / NaN inputs
    g2, _ = full_graph(mine, dev)
    win = 20

    def window(i):
        with torch.no_grad():
            xw = eng.philox_normal(g2.n_atoms * 119, 91 + i, 0xFFFFFFFF, 0, offset * 119, 1.0).view(-1, 119)
            pw = eng.philox_normal(g2.n_atoms * 3, 91 + i, 0xFFFFFFFF, 1, offset * 3, 1.0).view(-1, 3)
            eng.ddpm_sample(model, g2, xw, pw, None, 1.0, 1.0, n_steps=win, seed=91 + i, atom_offset=offset, options=plan)
        return xw, pw
    dtw, (xw, pw) = ctx.timed(window, 1, 3)
    out = {"workload": "configs[3]: 65-128-atom ligands, fully connected (i != j) edges, 1000-step DDPM, fp32, "
                       "molecules sharded by sum n^2, no collectives",
           "molecules": len(sizes), "molecules_rank0": len(mine), "atoms_rank0": graph.n_atoms,
           "edges_rank0": graph.n_edges, "ddpm_steps": T_STEPS, "seconds": dt,
           "molecules_per_s": len(sizes) / dt, "ms_per_reverse_step": dt / T_STEPS * 1e3,
           "finite": bool(torch.isfinite(x).all() and torch.isfinite(pos).all()),
           "finite_window": {"reverse_steps": win, "calls": 3, "ms_per_reverse_step": dtw / 3 / win * 1e3,
                             "molecules_per_s_at_1000_steps": len(sizes) / (dtw / 3 / win * T_STEPS),
                             "finite": bool(torch.isfinite(xw).all() and torch.isfinite(pw).all()),
                             "note": "timesteps 19..0 from fresh noise, includes each call's set-up (time tables, graph "
                                     "capture); the state stays finite over this window"},
           "kernel_forms": plan.as_dict(),
           "note": "TIMING ONLY for the 1000-step figure: with random-init weights a fully connected ligand of this size "
                   "leaves the finite range after a few dozen reverse steps (each atom's coordinate update sums over n - 1 "
                   "neighbours; the CPU oracle does the same step for step, tests/test_gpu_full_edges.py), so most of "
                   "the 1000 steps run on Inf / NaN inputs; finite_window is the same loop where the state is finite"}
    if ctx.rank == 0:      # dominant kernel at this shape: the fused edge path of an E(3) layer
        fc_traffic, fc_src = profile_traffic("fc")
        t_index = torch.full((1,), 500, dtype=torch.int32, device=dev)
        xr = torch.randn(g2.n_atoms, 119, device=dev)
        pr = torch.randn(g2.n_atoms, 3, device=dev)
        E, N = g2.n_edges, g2.n_atoms
        try:     # edge pre-activation, second edge linear + LN, coordinate gate (+ per-destination sums)
            eng.time_launch("e3_1.edge_chain", g2, xr, pr, t_index, reps=2, options=plan)
            ms = eng.time_launch("e3_1.edge_chain", g2, xr, pr, t_index, reps=5, options=plan)
            flops = 2.0 * E * (32 * 32 + 32 * 256 + 256 * 256 + 256 * 128)
            name = (f"edge_chain_kernel (e3_*.edge_chain, E={E}: radial 32x32 + K=32 + 256x256 + 256x128 per edge"
                    + (", per-destination message sums inside the kernel)" if plan.edge_reduce else ")"))
            needle = ("edge_chain_kernel<false, 0, true",) if plan.edge_reduce else ("edge_chain_kernel<false, 0, false",)
        except Exception:   # noqa: BLE001  (separate launches: the second edge-MLP GEMM is the largest)
            eng.time_launch("e3_1.edge_mlp2", g2, xr, pr, t_index, reps=2, options=plan)
            ms = eng.time_launch("e3_1.edge_mlp2", g2, xr, pr, t_index, reps=5, options=plan)
            flops = 2.0 * E * 256 * 256
            name = f"gemm_cs_kernel<APtrPlain, EpiLN> (e3_*.edge_mlp2, M={E} K=256 N=256)"
            needle = ("gemm_cs_kernel<", "EpiLN<0, 0, false>")
        ach = flops / (ms * 1e-3) / 1e12
        out["roofline"] = {"kernel": name, "bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": traffic_of(fc_traffic, *needle),
                           "traffic_source": fc_src, "avg_launch_ms": ms}
        ms_agg = eng.time_launch("e3_1.aggregate", g2, xr, pr, t_index, reps=5, options=plan)
        if plan.edge_reduce:
            rows = fc_partial_rows(mine)
            bytes_agg = rows * 1024.0 + N * 1024.0 + E * 32.0 + N * 36.0
            kname = (f"k_aggregate<true> (e3_*.aggregate on {rows} per-slab partial rows: the [E,256] messages are summed "
                     "per destination inside edge_chain_kernel and never reach memory)")
            needle = ("k_aggregate<true>",)
        else:
            bytes_agg = E * 1024.0 + N * 1024.0 + E * 32.0 + N * 36.0
            kname, needle = "k_aggregate", ("k_aggregate<false>",)
        out["roofline_aggregation"] = {"kernel": kname, "bound": "hbm", "achieved": bytes_agg / (ms_agg * 1e-3) / 1e9,
                                       "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                       "frac": bytes_agg / (ms_agg * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                       "traffic": traffic_of(fc_traffic, *needle), "traffic_source": fc_src,
                                       "algorithmic_bytes": bytes_agg, "avg_launch_ms": ms_agg,
                                       "message_tensor_bytes_avoided": 2.0 * E * 1024.0 if plan.edge_reduce else 0.0}
    return out


def leg_config5_ragged_split_sweep(ctx, mols_per_rank=1850):
    """BASELINE configs[4]: mixed-size ragged ligands (10-128 atoms, bond templates), 1000-step DDPM with
    the MLP GEMMs on the bf16 matrix pipe (split-operand mode); a bounded slice of the 100k-molecule
    sweep - 1850 molecules per rank, about one of the generator's split-mode batches (at most 131072 atoms) as
    the full sweep packs them - sharded by atoms.  The full sweep, run once on one
    GPU (tools/config5_sweep.py), is reported from its committed record."""
    import numpy as np
    from diffusion_model_ligandmol_generation_amd.generation import ResearchValidatedGenerator
    from diffusion_model_ligandmol_generation_amd.parallel import shard_ranges
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from config5_sweep import fingerprint
    dev = ctx.dev
    sizes = uniform_sizes(mols_per_rank * ctx.world, 10, 128, 555)
    lo, hi = shard_ranges(sizes, ctx.world, cost="atoms", align_atoms=32768)[ctx.rank]
    mine = sizes[lo:hi]
    model = build_model(dev, 128)
    gen = ResearchValidatedGenerator(model)
    eng = model.engine()
    gen.options = eng.options(gemm_mode=1)      # per call: the process-wide GEMM mode is left alone

    def one_pass(i):
        res, flags = gen.sample_many(mine, None, 1.0, 1.0, use_ddim=False, seed=31 + i, atom_offset=sum(sizes[:lo]))
        return res, flags
    dt, (res, flags) = ctx.timed(one_pass, 1, 1)      # one untimed pass first (allocations, graph capture)
    rate = len(sizes) / dt
    finite = [bool(np.isfinite(r[0]).all() and np.isfinite(r[1]).all()) for r in res]
    prints = {fingerprint(r[0], r[1]) for r, ok in zip(res, finite) if ok}
    out = {"workload": "configs[4]: ragged 10-128-atom template ligands, 1000-step DDPM, MLP GEMMs as split-operand bf16 "
                       f"MFMA (fp32 accumulate), batches of <= {gen.batch_atoms_split} atoms, results copied to the host",
           "molecules": len(sizes), "atoms": int(sum(sizes)), "ddpm_steps": T_STEPS, "seconds": dt,
           "molecules_per_s": rate, "extrapolated_100k_sweep_s": 100000.0 / rate,
           "accepted_rank0": int(sum(flags)), "acceptance_rate_rank0": float(sum(flags)) / max(1, len(mine)),
           "finite_rank0": int(sum(finite)), "finite": bool(all(finite)),
           "distinct_among_finite_rank0": len(prints) / max(1, sum(finite))}
    path = os.path.join(ROOT, "profiles", "r3_config5_sweep.json")
    if ctx.rank == 0 and os.path.exists(path):
        with open(path) as fh:
            out["full_sweep_recorded"] = dict(json.loads(fh.readline()), source="profiles/r3_config5_sweep.json "
                                              "(tools/config5_sweep.py, one MI355X, not re-run by this bench)")
    return out


def run_extras(ctx, extras):
    """Every leg on every rank (their barriers stay matched); a failing leg reports its error."""
    legs = (("config3_train_step", lambda: leg_config3_train_step(ctx)),
            ("config4_fc_sampling", lambda: leg_config4_fc_sampling(ctx)),
            ("config5_ragged_split_sweep", lambda: leg_config5_ragged_split_sweep(ctx)))
    for name, fn in legs:
        t0 = time.time()
        log(ctx, f"leg {name} ...")
        try:
            extras[name] = fn()
            extras[name]["leg_seconds"] = time.time() - t0
        except Exception as e:   # noqa: BLE001
            extras[name] = {"error": repr(e)}
        torch.cuda.empty_cache()


def run_rank(args):
    ctx = Ctx(args)
    world, rank, dev = ctx.world, ctx.rank, ctx.dev
    if args.gpus != world:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but {world} rank(s) joined", file=sys.stderr)
        sys.exit(2)

    from diffusion_model_ligandmol_generation_amd.generation import ResearchValidatedGenerator
    from diffusion_model_ligandmol_generation_amd.engine import template_graph
    model = build_model(dev)
    gen = ResearchValidatedGenerator(model)
    if args.one_call:
        gen.concurrent_atoms = None
    eng = model.engine()
    sizes = [ATOMS] * MOLS_PER_GPU
    atoms_per_rank = ATOMS * MOLS_PER_GPU

    def one_pass(i):
        # global molecule range of this rank in pass i -> Philox offsets independent of N
        offset = (i * world + rank) * atoms_per_rank
        with torch.no_grad():
            x, pos, graph, rejected = gen.sample_batch(sizes, None, 1.0, 1.0, seed=2024, atom_offset=offset)
            accept = eng.validity_gate(x, pos, graph.mol_ptr, graph.n_mols)
        return x, pos, int(accept.sum().item())

    pass_ends = []    # host clock after each timed pass of the last timed_run (a pass ends with the gate's .item())

    def timed_run(warmup, steps):
        accepted = [0]
        pass_ends.clear()

        def fn(i):
            if i == 0:
                pass_ends.append(time.time())
            x, pos, acc = one_pass(i)
            if i < 10_000:
                accepted[0] += acc
                pass_ends.append(time.time())
            return x, pos
        dt, (x, pos) = ctx.timed(fn, warmup, steps)
        return dt, accepted[0], x, pos

    # GEMM mode of the library: fp32 matrix pipe unless LIGDIFF_GEMM_SPLIT=1 is in the environment.
    # The headline run uses that mode; the other mode is timed afterwards and reported beside it.
    split_mode = os.environ.get("LIGDIFF_GEMM_SPLIT", "0")[:1] not in ("", "0")
    gen.options = eng.options(gemm_mode=int(split_mode))      # per call; no process-wide switch is touched
    log(ctx, f"headline: {args.warmup} warm-up + {args.steps} timed passes of {MOLS_PER_GPU} molecules x {T_STEPS} steps ...")
    dt, accepted, x, pos = timed_run(args.warmup, args.steps)
    pass_seconds = [round(b - a, 4) for a, b in zip(pass_ends, pass_ends[1:])]
    log(ctx, f"headline done: {dt:.2f} s; other GEMM mode ...")
    finite = bool(torch.isfinite(x).all() and torch.isfinite(pos).all())
    # uniqueness of the last pass's molecules on this rank (atom-type multiset + pairwise distances to 0.1)
    distinct = None
    if finite:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from config5_sweep import fingerprint
        xh, ph = x.detach().cpu().numpy(), pos.detach().cpu().numpy()
        prints = {fingerprint(xh[m * ATOMS:(m + 1) * ATOMS], ph[m * ATOMS:(m + 1) * ATOMS]) for m in range(MOLS_PER_GPU)}
        distinct = len(prints) / MOLS_PER_GPU
    # (every rank runs it, so the barriers inside stay matched; a failure here must not lose the headline)
    alt_error = None
    try:
        gen.options = eng.options(gemm_mode=int(not split_mode))
        dt_alt, _, x_alt, pos_alt = timed_run(1, args.steps)
    except Exception as e:   # noqa: BLE001
        alt_error = repr(e)
    finally:
        gen.options = eng.options(gemm_mode=int(split_mode))

    out = None
    log(ctx, "rooflines / CPU baselines ...")
    traffic, traffic_src = profile_traffic()
    split_traffic, split_src = profile_traffic("split")
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * MOLS_PER_GPU * args.steps / dt
        # ---- roofline of the dominant kernel (fp32 MFMA GEMM: FFN up-projection of a DiT layer)
        N = atoms_per_rank
        t_index = torch.full((1,), 500, dtype=torch.int32, device=dev)
        graph, _, _ = template_graph(sizes, dev)

        def ffn_up_roofline(split):
            ro = eng.plan_graph(graph, eng.options(gemm_mode=int(split)))
            if not split:
                # fp32 pipe: the dominant launch is the fused attention-layer tail (out-projection, feed-forward
                # block, next layer's pre-norm and in-projection); LIGDIFF_CHAIN=0 brings the separate
                # launches back and the feed-forward up-projection is then the largest one
                try:
                    eng.time_launch("dit1.chain", graph, x, pos, t_index, reps=2, options=ro)    # warm
                    ms = eng.time_launch("dit1.chain", graph, x, pos, t_index, reps=10, options=ro)
                    # 29-atom ligands take the molecule-per-wave form: the layer's own in-projection and attention
                    # (8 heads x (K Q^T + V^T P^T) on 32 x 32 x 32 tiles per molecule) sit in front of the out-projection
                    mol_form = ATOMS <= 32 and os.environ.get("LIGDIFF_CHAIN_MOL", "1")[:1] != "0"
                    flops = 2.0 * N * (256 * 256 + 2 * 256 * 1024 + 256 * 768)
                    if mol_form:
                        flops += MOLS_PER_GPU * 8 * 2 * 2.0 * 32 * 32 * 32
                        name = ("dit_chain_kernel<DIT_TAIL_NONE, MOLWAVE> (dit*.chain: pre-norm, in-projection 256-768, "
                                "attention over the molecule, out-projection + gated residual + LN, FFN 256-1024-256 with GELU, "
                                f"gated residual; one molecule per wave, M={N})")
                        needle = "dit_chain_kernel<0, true"
                    else:
                        name = ("dit_chain_kernel<DIT_TAIL_QKV> (dit*.chain: out-projection + gated residual + LN, FFN "
                                "256-1024-256 with GELU, gated residual + next layer's LN and in-projection 256-768; "
                                f"M={N})")
                        needle = "dit_chain_kernel<1, false"
                    ach = flops / (ms * 1e-3) / 1e12
                    return {"kernel": name,
                            "bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                            "frac": ach / PEAK_F32_MFMA_TFLOPS, "flops_per_launch": flops,
                            "traffic": traffic_of(traffic, needle),
                            "traffic_source": traffic_src, "avg_launch_ms": ms}
                except Exception:   # noqa: BLE001  (label not present: chain kernels switched off)
                    pass
            eng.time_launch("dit1.ffn_up", graph, x, pos, t_index, reps=2, options=ro)           # warm
            ms = eng.time_launch("dit1.ffn_up", graph, x, pos, t_index, reps=10, options=ro)
            flops = 2.0 * N * 256 * 1024
            if not split:
                ach = flops / (ms * 1e-3) / 1e12
                return {"kernel": "gemm_cs_kernel<APtrPlain, EpiBiasAct<8,GELU>> (dit*.ffn_up, M=29696 K=256 N=1024)",
                        "bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": ach / PEAK_F32_MFMA_TFLOPS,
                        "traffic": traffic_of(traffic, "gemm_cs_kernel<", "EpiBiasAct<8, 1, false, false>"),
                        "traffic_source": traffic_src, "avg_launch_ms": ms}
            # six bf16 x bf16 products per fp32 product: the matrix pipe executes 6x the algorithmic flops
            ach = 6.0 * flops / (ms * 1e-3) / 1e12
            return {"kernel": "gemm_cs_split_kernel<APtrPlain, EpiBiasAct<8,GELU>> (dit*.ffn_up, M=29696 K=256 N=1024)",
                    "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / PEAK_BF16_MFMA_TFLOPS,
                    "traffic": traffic_of(split_traffic, "gemm_cs_split_kernel<", "EpiBiasAct<8, 1, false"),
                    "traffic_source": split_src, "avg_launch_ms": ms,
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
                        "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": agg_gbs / PEAK_HBM_GBS,
                        "traffic": traffic_of(traffic, "k_aggregate"), "traffic_source": traffic_src,
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
                       "parallelism": f"molecule-sharded x{world}, no collectives",
                       "streams_per_gpu": ("1 (--one-call)" if args.one_call else
                                           "2 (the batch runs as two concurrent halves of 512 molecules; same bits as one call)")},
            "pass_seconds_rank0": pass_seconds,
            "attempted_molecules": world * MOLS_PER_GPU * args.steps,
            "accepted_rank0": accepted, "finite": finite, "distinct_last_pass_rank0": distinct,
            "roofline": roofline, "roofline_aggregation": roofline_agg,
            # the same K passes in the library's other GEMM mode (not the headline; see config.gemm_mode)
            "alt_gemm_mode": alt,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            out["cpu_baseline_batched"] = cpu_baseline_batched()
            try:   # BASELINE configs[0]: the main.py flow at batch 64 (1 train step + 100-step DDPM), oracle beside it
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import run_config1
                out["config1"] = run_config1.run(device=dev)
            except Exception as e:   # noqa: BLE001
                out["config1"] = {"error": repr(e)}

    # ---- the other named workloads.  The headline above is final; a watchdog prints it alone if the legs
    # below do not come back in time (e.g. one rank stuck in a collective), so that it cannot be lost.
    printed = threading.Lock()
    extras = {}

    def emit(note=None):
        if not printed.acquire(blocking=False):
            return
        if rank == 0:
            line = dict(out)
            line.update(extras)
            if note:
                line["extras_note"] = note
            print(json.dumps(line), flush=True)

    def watchdog():
        # a stuck leg or collective: the measured headline still goes out, but the run is reported as failed
        emit(f"extra legs abandoned after {EXTRAS_BUDGET_S:.0f} s")
        sys.stdout.flush()
        print(f"bench.py: extra legs did not return within {EXTRAS_BUDGET_S:.0f} s; exiting with status 3", file=sys.stderr,
              flush=True)
        os._exit(3)

    if args.no_extras:
        emit()
    else:
        timer = threading.Timer(EXTRAS_BUDGET_S, watchdog)
        timer.daemon = True
        timer.start()
        ctx.fence()
        run_extras(ctx, extras)
        timer.cancel()
        log(ctx, "extras done")
        emit()
    if ctx.dist is not None:
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only (profiling runs)")
    ap.add_argument("--one-call", action="store_true",
                    help="run each batch as ONE library call on one stream instead of two concurrent halves "
                         "(profiling runs: per-kernel durations then belong to whole-batch launches)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    if os.environ.get("LIGDIFF_BENCH_TRACE"):      # debugging aid: Python stacks on stderr every N seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["LIGDIFF_BENCH_TRACE"]), repeat=True, file=sys.stderr)
    run_rank(args)


if __name__ == "__main__":
    main()
