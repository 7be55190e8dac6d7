#!/usr/bin/env python
"""bench.py — throughput of the accelerated path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mols 512] [--kind qm9] [--mode fwd|train]

A *step* is one pass of the hot path — ``BondMessagePassing.forward`` (graph plan K0 + K1..K5,
chemprop/nn/message_passing/base.py:196-212) — over one batch of synthetic QM9-shaped molecules that
is already resident in HBM (BASELINE.json configs[1]: depth 3, hidden 300, batch 512 molecules per
GPU).  ``--mode train`` adds the backward pass (K6) and, for N > 1, the RCCL gradient all-reduce.
The metric is BASELINE.json's: million directed-edge-updates / s = E * (depth - 1) / t.

For N > 1 the driver launches one process per GPU (torch.distributed.run); molecules are sharded
across ranks (every rank owns its own batch: weak scaling) and the forward needs no collective.
``value`` is the SAME metric at every N (the forward, so the per-N values of a scaling sweep can be
divided by each other); the step that does hold a collective — forward + backward + ONE RCCL
all-reduce of the flat gradient buffer — is timed on all ranks as well and reported as ``train_step``
in the same line, at every N.

Rank 0 prints ONE JSON line, with extra objects:
  roofline      dominant kernel (the whole-forward tile kernel on the f16 pipe) timed live with HIP
                events on the launch stream: achieved = algorithmic FLOP / mean launch time
  cpu_baseline  the EXECUTED reference (chemprop.nn.BondMessagePassing staged under oracle/_ref, kind "reference"; the
                restated op sequence, kind "port", only where no reference tree travelled) timed on the host cores at
                {1, 8, 16, 32, all} threads, bounded to ~20 s; N = 1 only; cpu_baseline_train: forward + backward
  train_step    forward (kept tensors) + backward (+ all-reduce at N > 1) of the same shard
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TF = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_HBM_GBS = 8000.0      # spec; 6290 GB/s measured copy ceiling (same guide)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mols", type=int, default=512, help="molecules per GPU per step")
    ap.add_argument("--kind", default="qm9", choices=["qm9", "zinc", "synth40", "cgr"])
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--hidden", type=int, default=300)
    ap.add_argument("--mode", default="fwd", choices=["fwd", "train"],
                    help="what `value` times.  fwd (default at EVERY N, so that the per-N values of one scaling sweep are the same "
                         "metric — BASELINE's forward edge-updates/s; molecule shards, no data-path collective); at N > 1 the same "
                         "line also carries `train_step`: forward + backward + the RCCL gradient all-reduce of BASELINE configs[3], "
                         "timed on all ranks with the same barrier / max-over-ranks rule.  train: `value` is that step instead")
    ap.add_argument("--groups", type=int, default=5, help="timed regions of exactly --steps steps each; `value` is their median")
    ap.add_argument("--no-graph", action="store_true", help="time eager launches only (no hipGraph replay)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large-batches", action="store_true",
                    help="skip the side measurements at 4096 / 32768 molecules (profiler runs: keeps every kernel's launches at the headline size)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    return ap.parse_args()


def time_events(fn, reps, torch):
    """Device time of ``fn`` in ms per call, HIP events on the current (= launch) stream: the reps are timed in five
    groups and the fastest group's mean is returned (one allocator growth or clock dip does not poison the figure)."""
    groups = 5 if reps >= 10 else 1
    per = max(reps // groups, 1)
    best = None
    for _ in range(groups):
        start = torch.cuda.Event(enable_timing=True)
        stop = torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(per):
            fn()
        stop.record()
        stop.synchronize()
        t = start.elapsed_time(stop) / per
        best = t if best is None else min(best, t)
    return best


def compact_line(out):
    """The final JSON line: the contract's keys as they are, every side measurement reduced to its numbers."""
    def pick(d, *keys):
        return {k: d[k] for k in keys if isinstance(d, dict) and k in d}

    c = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                             "vs_baseline") if k in out}
    c["dtype"] = "f32" if out.get("dtype") == "f32" else "f32 (exact 3-term f16 split of fp32 operands, fp32 accumulate)"
    c["data"] = out.get("data")
    cfg = out.get("config", {})
    c["config"] = pick(cfg, "workload", "mols_per_gpu", "directed_edges_per_gpu", "parallelism", "launch")
    r = out.get("roofline", {})
    c["roofline"] = pick(r, "bound", "achieved", "peak", "unit", "frac", "traffic", "launch_us", "frac_of_fp32_mfma_peak", "mfma_busy")
    if "kernel" in r:
        c["roofline"]["kernel"] = str(r["kernel"]).split(":")[0].split(" (")[0]
    cb = out.get("cpu_baseline", {})
    if cb:
        c["cpu_baseline"] = pick(cb, "value", "unit", "cores", "kind", "ms_per_step")
        c["cpu_baseline"]["sample"] = str(cb.get("sample", "")).split(";")[0][:120]
    for k in ("speedup_vs_cpu", "eager_ms_per_step", "graph_ms_per_step", "graph_k_steps_ms_per_step", "edges_per_s_M", "route"):
        if k in out:
            c[k] = out[k]
    for k in ("roofline_scatter", "roofline_scatter_large", "roofline_per_step_kernel", "roofline_large", "roofline_steps16_large"):
        if k in out:
            c[k] = pick(out[k], "frac", "achieved", "unit", "launch_us", "error")
    if "host_handoff" in out:
        c["host_handoff"] = pick(out["host_handoff"], "five_tensors_us", "packed_us", "error")
    if "half_operands" in out:
        c["half_operands"] = pick(out["half_operands"], "us", "M_edge_updates_per_s", "route", "max_diff_vs_exact_over_max_out", "error")
    if "loader_tiles" in out:
        c["loader_tiles"] = {k: pick(v, "loader_tiles_us", "device_plan_us") for k, v in out["loader_tiles"].items() if isinstance(v, dict)}
    if "cpu_baseline_train" in out:
        c["cpu_baseline_train"] = pick(out["cpu_baseline_train"], "value", "cores", "kind", "ms_per_step")
    for k in ("train_speedup_vs_cpu", "rccl"):
        if k in out:
            c[k] = out[k]
    if "train_step_dropout" in out:
        c["train_step_dropout"] = pick(out["train_step_dropout"], "fused_us", "rows_route_us", "error")
    if "kernel_breakdown" in out:
        c["kernel_breakdown"] = pick(out["kernel_breakdown"], "tile_plan_us", "plan_us", "error")
        if "launch_us" in r:
            c["kernel_breakdown"]["tile_kernel_us"] = r["launch_us"]
    if "other_configs" in out:
        oc = {}
        for name, v in out["other_configs"].items():
            short = name.split(" (")[0]
            e = pick(v, "us", "route", "train_step_us", "f16_storage_us", "error")
            if isinstance(v, dict) and "roofline" in v:
                e["frac_mfma"] = v["roofline"].get("frac_mfma")
            oc[short] = e
        c["other_configs"] = oc
    if "train_step" in out:
        c["train_step"] = pick(out["train_step"], "ms_per_step", "M_edge_updates_per_s", "allreduce_exposed_us", "collectives_launched", "error")
    if "model_step" in out:
        c["model_step"] = pick(out["model_step"], "fused_ms_per_step", "module_path_ms_per_step", "fused_M_edge_updates_per_s", "route", "error")
    c["detail"] = "the BENCH_DETAIL line above: every note, sample description and per-config roofline block"
    return c


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    from chemprop_amd import _lib, engine, synth
    from chemprop_amd.nn import BondMessagePassing

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...`")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
    # (DMPNN_BENCH_BACKEND=gloo DMPNN_BENCH_DEVICE=0: a dry run of the N > 1 control flow on a one-GPU box — every rank on the same
    #  device, collectives through the host; the numbers of such a run mean nothing)
    backend = os.environ.get("DMPNN_BENCH_BACKEND", "nccl")
    dev = torch.device("cuda", int(os.environ.get("DMPNN_BENCH_DEVICE", local_rank)))
    torch.cuda.set_device(dev)
    # (launched by torchrun — RANK / MASTER_ADDR in the environment — the process group is initialised also at world size 1: on a
    #  one-GPU box `torchrun --nproc-per-node 1 bench.py --gpus 1` then runs the barriers, the MAX all-reduce of the timing and, with
    #  DMPNN_FORCE_COLLECTIVE=1, the gradient all-reduce of the training step on RCCL itself: profiles/r05_nccl_world1.json)
    use_pg = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if use_pg:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    _lib.load()

    # ---- workload: this rank's shard (molecules hash-partitioned by id -> independent batches) ----
    bmg = synth.random_batch(args.mols, args.kind, seed=1000 + rank)
    d_v, d_e = int(bmg.V.shape[1]), int(bmg.E.shape[1])
    torch.manual_seed(0)
    mp = BondMessagePassing(d_v=d_v, d_e=d_e, d_h=args.hidden, depth=args.depth)
    cpu_state = {k: v.clone() for k, v in mp.state_dict().items()}
    mp = mp.to(dev)
    cpu_bmg = synth.random_batch(args.mols, args.kind, seed=1000 + rank) if rank == 0 else None
    bmg.to(dev)
    nV, nE = int(bmg.V.shape[0]), int(bmg.E.shape[0])
    updates = nE * (args.depth - 1)

    train = args.mode == "train"
    if train:
        from chemprop_amd import distributed as ddp

        mp.train()
        params = [p for p in mp.parameters()]
        G = torch.randn(nV, mp.output_dim, device=dev)

        # one flat gradient buffer the backward kernels write into; ONE all-reduce per step, launched on a communication
        # stream right behind the backward pass (chemprop_amd/distributed.py: GradSync)
        from chemprop_amd.optim import FlatAdam

        sync = ddp.GradSync(params, modules=[mp])
        opt = FlatAdam(sync, lr=1e-4)  # (chemprop trains with Adam, models/model.py:208-231: one fused launch over the flat buffers)

        def step():
            with ddp.backward_on_calling_thread():  # (one process per GPU: no hand-off to autograd's device thread, ~95 µs of host time)
                out = mp(bmg)   # (after the module's first validated batches K0 is the 11 us tile table: DMPNN_F_TILE_PLAN)
                out.backward(G)
            sync.allreduce()
            opt.step()             # (waits for the exchange on the stream, folds in 1 / world, updates: nothing is skipped)
    else:
        mp.eval()

        def step():
            with torch.no_grad():
                return mp(bmg)

    def run_steps(fn, n):
        for _ in range(n):
            fn()

    def timed(fn, n):
        """EXACTLY n steps between a barrier + device synchronize on both sides; the clock is read before the closing
        barrier (a collective is not part of the timed region); the maximum over the ranks is what counts."""
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(fn, n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if use_pg:
            dist.barrier()
        if use_pg:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # ---- eager: W warm-up steps, then exactly K timed steps ----
    run_steps(step, args.warmup)
    # `value` comes from the MEDIAN of --groups (default 5) timed regions of exactly --steps steps each (every region bracketed by
    # barrier + synchronize, maximum over the ranks): one 1 ms interval on a shared host is not a headline
    def timed_groups(fn, n, groups):
        ts = sorted(timed(fn, n) for _ in range(max(groups, 1)))
        return ts[len(ts) // 2], ts

    eager_s, eager_all = timed_groups(step, args.steps, args.groups)
    eager_ms = eager_s / args.steps * 1e3

    # ---- hipGraph replay of the same step (launch-bound regime: 512 molecules ~ 10 short kernels) ----
    graph_ms, graph_err = None, None
    if not args.no_graph and not train:  # (the training step is timed eagerly: its optimizer step takes per-step host scalars)
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                run_steps(step, 3)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                step()
            run_steps(g.replay, args.warmup)
            graph_ms = timed_groups(g.replay, args.steps, args.groups)[0] / args.steps * 1e3
        except Exception as e:  # report, never hide
            graph_err = f"{type(e).__name__}: {e}"[:300]
    # ---- the same K steps captured as ONE hipGraph (K x (K0 + forward) nodes, one launch per timed region): what a serving loop that
    # knows its next K batches would replay; the region loses K - 1 launch calls and most of its pipeline fill ----
    graphk_ms, graphk_err = None, None
    if graph_ms is not None and args.steps <= 512:
        try:
            gk = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gk):
                run_steps(step, args.steps)
            run_steps(gk.replay, 2)
            graphk_ms = timed_groups(gk.replay, 1, args.groups)[0] / args.steps * 1e3
            del gk
        except Exception as e:  # report, never hide
            graphk_err = f"{type(e).__name__}: {e}"[:300]
    cands = [("eager", eager_ms)] + ([("hipGraph replay", graph_ms)] if graph_ms is not None else []) + \
            ([(f"hipGraph replay, {args.steps} steps per launch", graphk_ms)] if graphk_ms is not None else [])
    # `value` is the EAGER figure: the mode forward() runs in behind chemprop's Lightning loop (one unknown batch at a time); the
    # hipGraph replays of the same step are side figures (graph_ms_per_step, graph_k_steps_ms_per_step) — a serving loop's business
    launch_name, ms_per_step = "eager", eager_ms
    best_name, best_ms = min(cands, key=lambda c: c[1])
    value = world * updates / (ms_per_step * 1e-3) / 1e6
    if train:
        sync.wait()

    out = {
        "metric": "million directed-edge-updates/sec (depth=%d, hidden=%d)" % (args.depth, args.hidden),
        "value": round(value, 3), "unit": "M edge-updates/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (contractions: 3-term exact f16 split of the fp32 operands, fp32 accumulate; fp32-MFMA build: DMPNN_MFMA=f32)"
                 if os.environ.get("DMPNN_MFMA", "split16") != "f32" else "f32", "data": "synthetic",
        "config": {"workload": f"{args.kind}-shaped synthetic molecules, {args.mols} mols/GPU/step, "
                               f"BondMessagePassing depth={args.depth} hidden={args.hidden}, "
                               + ("forward+backward" + ("+RCCL grad all-reduce" if world > 1 else "") + "+fused Adam step" if train else "forward (plan K0 + K1..K5)"),
                   "mols_per_gpu": args.mols, "atoms_per_gpu": nV, "directed_edges_per_gpu": nE,
                   "parallelism": f"dp{world} (molecule shards, no data-path collective)",
                   "launch": launch_name},
        "timing": {"groups": args.groups, "steps_per_group": args.steps, "statistic": "median group",
                   "eager_ms_per_step_by_group": [round(t / args.steps * 1e3, 5) for t in eager_all]},
        "rccl": {"world_size": (dist.get_world_size() if use_pg else 1), "backend": (dist.get_backend() if use_pg else None),
                 "forced_collective_at_world_1": bool(use_pg and world == 1 and os.environ.get("DMPNN_FORCE_COLLECTIVE") == "1")},
        "parity_bar": {"forward": "<= 1e-5 norm-wise against the executed reference (tests/)",
                       "gradients": "<= 2e-5 norm-wise against the executed reference's autograd (a kinked activation's gradient is only as "
                                    "reproducible as its masks: DESIGN.md section 5)"},
        "eager_ms_per_step": round(eager_ms, 5),
        "best_launch_mode": {"launch": best_name, "ms_per_step": round(best_ms, 5),
                             "M_edge_updates_per_s": round(world * updates / (best_ms * 1e-3) / 1e6, 3)},
        "graph_ms_per_step": None if graph_ms is None else round(graph_ms, 5),
        "graph_k_steps_ms_per_step": None if graphk_ms is None else round(graphk_ms, 5),
        "edges_per_s_M": round(world * nE / (ms_per_step * 1e-3) / 1e6, 3),
    }
    out["weights"] = ("the pre-split (hi + lo f16) of W_i / W_h / W_o is redone by EVERY step — it rides in K0's launch (workgroup 0 plans, "
                      "the others split): nothing about the weights is cached between forwards (round 3 kept it keyed on autograd versions)")
    if graph_err:
        out["graph_error"] = graph_err
    if graphk_err:
        out["graph_k_steps_error"] = graphk_err

    # ---- the training step of the same shard: forward with kept tensors + backward into the flat gradient buffer + (N > 1) ONE
    # RCCL all-reduce on the communication stream (chemprop_amd/distributed.py: GradSync).  On ALL ranks, same timing rule as
    # `value` (barrier + synchronize on both sides, maximum over the ranks) ----
    if not train:
        try:
            from chemprop_amd import distributed as ddp

            tmp = BondMessagePassing(d_v=d_v, d_e=d_e, d_h=args.hidden, depth=args.depth).to(dev).train()
            tmp.load_state_dict(mp.state_dict())
            tps = list(tmp.parameters())
            from chemprop_amd.optim import FlatAdam

            tsync = ddp.GradSync(tps, modules=[tmp])
            topt = FlatAdam(tsync, lr=1e-4)      # (chemprop trains with Adam, models/model.py:208-231: one fused launch here)
            Gt = torch.randn(nV, tmp.output_dim, device=dev)

            def tstep():
                with ddp.backward_on_calling_thread():  # (see chemprop_amd/distributed.py: the step was host-bound without it)
                    o = tmp(bmg)
                    o.backward(Gt)
                tsync.allreduce()
                topt.step()                      # (waits for the exchange on the stream, divides by the world size, updates)

            run_steps(tstep, 10)
            t_tr = timed_groups(tstep, args.steps, args.groups)[0] / args.steps * 1e3  # (the rule of `value`: groups of K steps, synchronize on both sides, max over ranks, median group)
            tsync.wait()
            exposed_us = None
            if world > 1:
                # what the exchange COSTS the step: the same step with the all-reduce left out (every rank updates with its own
                # gradients: numerically another run, the same kernels), same timing rule — the difference is the part of the
                # collective the backward pass does not hide
                def tstep_noex():
                    with ddp.backward_on_calling_thread():
                        o = tmp(bmg)
                        o.backward(Gt)
                    topt.step()
                run_steps(tstep_noex, 5)
                t_no = timed_groups(tstep_noex, args.steps, args.groups)[0] / args.steps * 1e3
                exposed_us = round((t_tr - t_no) * 1e3, 1)
                run_steps(tstep, 2)
                tsync.wait()
            out["train_step"] = {"ms_per_step": round(t_tr, 5), "M_edge_updates_per_s": round(world * updates / (t_tr * 1e-3) / 1e6, 2),
                                 "allreduce_exposed_us": exposed_us,
                                 "n_gpus": world, "collective": "one RCCL all-reduce of the flat gradient buffer per step" if (world > 1 or (use_pg and os.environ.get("DMPNN_FORCE_COLLECTIVE") == "1")) else None,
                                 "collectives_launched": int(tsync.n_collectives),
                                 "autograd": "backward on the calling thread (torch.autograd.set_multithreading_enabled(False): one process per GPU)",
                                 "plan": "K0 inside every step, on the stream: the tile table (dmpnn_prepare_tiles, 11 us; the kept tensors stay in the "
                                         "caller's edge order, DMPNN_F_TILE_PLAN)",
                                 "note": "forward (kept tensors) + backward + gradient exchange + fused Adam step of the block's parameters, same "
                                         "shard, eager; weak scaling of THIS figure is the data-parallel training claim (BASELINE configs[3])"}
            del tmp, tsync
        except Exception as e:
            out["train_step"] = {"error": f"{type(e).__name__}: {e}"[:200]}

    # ---- f4: the WHOLE model's training step (models/model.py:148-161 + Adam): block + aggregation + batch norm + predictor +
    # loss + backward of all of it + optimizer.  (a) the module path: the same kernels driven from Python through autograd, torch
    # modules for batch norm / criterion, torch.optim.Adam;  (b) FusedTrainer: ONE dmpnn_train_step call.  N = 1 only. ----
    if not train and world == 1 and args.hidden == 300:
        try:
            from chemprop_amd import agg as cagg
            from chemprop_amd import distributed as ddp
            from chemprop_amd.model import MPNN, FusedTrainer, RegressionFFN

            def make_model():
                torch.manual_seed(0)
                return MPNN(BondMessagePassing(d_v=d_v, d_e=d_e, d_h=args.hidden, depth=args.depth), cagg.NormAggregation(),
                            RegressionFFN(n_tasks=1, input_dim=args.hidden), batch_norm=True).to(dev).train()

            from chemprop_amd.optim import FlatAdam as FlatAdamM

            y = torch.randn(args.mols, 1, device=dev)
            m_a = make_model()
            sync_a = ddp.GradSync(list(m_a.parameters()), modules=[m_a])
            opt_a = FlatAdamM(sync_a, lr=1e-4)

            def step_module():   # (what integration.HipMPNN.training_step runs for a model the fused step refuses)
                with ddp.backward_on_calling_thread():
                    sync_a.zero_grad()
                    m_a.loss(bmg, y).backward()
                sync_a.allreduce()
                opt_a.step()

            m_b = make_model()
            tr_b = FusedTrainer(m_b, lr=1e-4)

            def step_fused():
                tr_b.step(bmg, y)

            run_steps(step_module, 10)
            t_mod = timed_groups(step_module, args.steps, args.groups)[0] / args.steps * 1e3
            run_steps(step_fused, 10)
            t_fus = timed_groups(step_fused, args.steps, args.groups)[0] / args.steps * 1e3
            out["model_step"] = {"fused_ms_per_step": round(t_fus, 5),
                                 "module_path_ms_per_step": round(t_mod, 5),
                                 "fused_M_edge_updates_per_s": round(updates / (t_fus * 1e-3) / 1e6, 2),
                                 "route": tr_b.last_route,
                                 "model": f"MPNN(BondMessagePassing(d_h={args.hidden}, depth={args.depth}), NormAggregation, BatchNorm1d, "
                                          "RegressionFFN(1 task, hidden 300), MSE) + Adam",
                                 "plan": "K0 inside the step's C call, on the critical path (the tile table, 11 us)",
                                 "head": "4 launches (round 5): k_agg_bn_fwd (aggregation + batch norm + the hidden layer's weight split), "
                                         "k_head_rows<., 1 | 2> (predictor + criterion + their backward over row block x column slice, f16 pipe), "
                                         "k_bn_agg_bwd; DMPNN_HEAD=chain: the 9 launches of rounds 3-4",
                                 "note": "fused: ONE C call (dmpnn_train_step) enqueues K0, the block's forward, aggregation, batch norm, the "
                                         "predictor, the loss, the backward pass of all of it and the Adam update; module path (round 4): "
                                         "MPNN.loss(batch).backward() through torch autograd — TWO nodes, the block (dmpnn_forward / "
                                         "dmpnn_backward) and everything behind it (dmpnn_head: aggregation, batch norm, predictor, criterion and "
                                         "their backward in one call) — then the flat Adam (one launch); round 3 timed torch's batch norm / loss "
                                         "ops and torch.optim.Adam there (0.80 ms)"}
            del m_a, m_b, tr_b, opt_a, sync_a
        except Exception as e:
            out["model_step"] = {"error": f"{type(e).__name__}: {e}"[:300]}

    # ---- the block's training step with ACTIVE dropout (CLI --dropout; hpopt searches it): inside the tile kernels (hash mask,
    # round 3) against the rows route (the block's own nn.Dropout between row kernels chained from Python: what every round before
    # did, and what a dropout module that is not exactly nn.Dropout still gets) ----
    if not train and world == 1 and args.hidden == 300 and not args.no_large_batches:
        try:
            from chemprop_amd import distributed as ddp
            from chemprop_amd.optim import FlatAdam

            class _RowsDropout(torch.nn.Dropout):  # (a subclass keeps its own semantics: the engine leaves it to torch)
                pass

            res = {}
            Gd = torch.randn(nV, args.hidden, device=dev)
            for tag, drop_cls in (("fused_us", None), ("rows_route_us", _RowsDropout)):
                torch.manual_seed(0)
                md = BondMessagePassing(d_v=d_v, d_e=d_e, d_h=args.hidden, depth=args.depth, dropout=0.2).to(dev).train()
                if drop_cls is not None:
                    md.dropout = drop_cls(0.2)
                sd = ddp.GradSync(list(md.parameters()), modules=[md])
                od = FlatAdam(sd, lr=1e-4)

                def dstep():
                    with ddp.backward_on_calling_thread():
                        md(bmg).backward(Gd)
                    sd.allreduce()
                    od.step()
                run_steps(dstep, 10)
                res[tag] = round(timed_groups(dstep, args.steps, args.groups)[0] / args.steps * 1e6, 1)
                del md, sd, od
            res["p"] = 0.2
            out["train_step_dropout"] = res
        except Exception as e:
            out["train_step_dropout"] = {"error": f"{type(e).__name__}: {e}"[:200]}

    if rank == 0:
        # ---- roofline of the dominant kernel, live HIP-event timing on the launch stream ----
        h = args.hidden
        plan = engine.GraphPlan.from_bmg(bmg)
        Wh = mp.W_h.weight.detach()
        fusable = h % 4 == 0 and h <= 320
        fwd_flop = 2.0 * nE * (d_v + d_e) * h + 2.0 * nE * h * h * (args.depth - 1) + 2.0 * nV * (d_v + h) * h
        route_used = None
        try:
            with torch.no_grad():
                _, st0 = engine.forward(plan, bmg.V, bmg.E, mp.W_i.weight, mp.W_h.weight, mp.W_o.weight, mp.W_o.bias,
                                        depth=args.depth)
            route_used = st0.route
        except Exception as e:
            out["route_error"] = f"{type(e).__name__}: {e}"[:200]
        out["route"] = route_used
        if route_used in ("mega16", "mega"):
            # ONE launch = the whole forward of every tile of whole molecules (k_mpnn_tile16 / k_mpnn_tile)
            # (the tile kernel ALONE: the argument block of one forward, replayed with DMPNN_F_WSPLIT_READY — its workspace holds the
            #  pre-split of these very weights; K0 and the weight pre-split are the step's other launch and are not part of this figure)
            import ctypes as _C

            # (... as the timed step launches it: on the TILE plan K0 builds from the batch vector — rows in the caller's order, no CSR
            #  arrays — and with the batch's molecule count as the launch's workgroup bound, chemprop_amd/nn.py: _replay_forward)
            plan_dom = engine.GraphPlan.from_bmg(bmg, light="tiles") if route_used == "mega16" else plan
            with torch.no_grad():
                _, st_dom = engine.forward(plan_dom, bmg.V, bmg.E, mp.W_i.weight, mp.W_h.weight, mp.W_o.weight, mp.W_o.bias, depth=args.depth)
            blk_dom = _lib.FwdArgs.from_buffer_copy(bytes(st_dom.args))
            blk_dom.flags |= _lib.F_WSPLIT_READY
            if route_used == "mega16" and plan_dom.tiles_only:
                blk_dom.n_tiles_launch = args.mols
            lib_dom = _lib.load()

            ref_dom, stream_dom = _C.byref(blk_dom), engine._stream_ptr(dev)   # (hoisted: the loop must stay shorter on the host than on the device)

            def kdom():
                if lib_dom.dmpnn_forward(ref_dom, stream_dom):
                    _lib.check(1, "dmpnn_forward")
            run_steps(kdom, 10)
            t_dom = time_events(kdom, 200, torch)   # (five groups of 40 back-to-back launches: the first launch of a group starts behind an idle queue)
            if route_used == "mega16":
                waves = int(lib_dom.dmpnn_tile_waves(nV, nE, h, 0))
                kname = (f"k_mpnn_tile16<5, {waves} waves>: whole forward per tile of whole molecules in one launch"
                         + (" (one 512-thread workgroup per tile, column tiles 3+3+3+3+2+2+2+2)" if waves == 8 else "") + "; "
                         "contractions as 3 x v_mfma_f32_16x16x32_f16 on exactly split fp32 operands (x s = hi + lo), fp32 accumulate")
                peak, peak_note = 2500.0 / 3.0, "f16 MFMA dense peak 2.5 PF / 3 MFMA passes per fp32 product"
            else:
                kname = "k_mpnn_tile<5>: whole forward per tile in one launch, exact fp32 MFMA 16x16x4"
                peak, peak_note = PEAK_FP32_MFMA_TF, "fp32 MFMA (= vector) peak"
            achieved = fwd_flop / (t_dom * 1e-3) / 1e12
            # algorithmic bytes of the launch: features + indices in, output out, weights once (they stream from L2 per tile)
            bytes_dom = 4.0 * (nV * d_v + nE * d_e + nV * h) + 12.0 * nE + 4.0 * h * ((d_v + d_e) + h + (d_v + h))
            out["roofline"] = {"kernel": kname, "bound": "mfma", "achieved": round(achieved, 3), "peak": round(peak, 1),
                               "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": None, "peak_note": peak_note,
                               "launch_us": round(t_dom * 1e3, 3), "flop_per_launch": fwd_flop,
                               "algorithmic_bytes_per_launch": bytes_dom,
                               "frac_of_fp32_mfma_peak": round(achieved / PEAK_FP32_MFMA_TF, 4),
                               "note": "one launch = the whole forward given the plan and the pre-split weights; H / M never leave the CU in this route"}
        Mbuf = torch.randn(nE, h, device=dev)
        H0buf = torch.randn(nE, h, device=dev)
        Mnext = torch.empty(nE, h, device=dev)
        if fusable:
            k3 = lambda: engine.update_fused(plan, Mbuf, H0buf, Wh, None, act="relu", want_M=True, M_next=Mnext)
            kname = "k_gemm<3,5,4,false,EPI_SEG> (per-step fused update: H'=relu(H0+M@W_h^T); M_next[rev]=S[dst]-H', fp32 MFMA 16x16x4)"
        else:
            Cbuf = torch.empty(nE, h, device=dev)
            k3 = lambda: engine.linear(Mbuf, Wh, None, Cadd=H0buf, act="relu", out=Cbuf)
            kname = "k_gemm<EPI_PLAIN> (K3 update: H = relu(H0 + M @ W_h^T), fp32 MFMA 16x16x4)"
        run_steps(k3, 10)
        t_k3 = time_events(k3, 50, torch)
        flops = 2.0 * nE * h * h
        bytes_k3 = 3.0 * nE * h * 4 + 3.0 * nE * 4  # SURVEY §8d B_upd: read M, read H0, write M_next (+ indices)
        achieved = flops / (t_k3 * 1e-3) / 1e12
        traffic = None
        try:  # PMC-derived HBM bytes per launch, produced by scripts/pmc_traffic.py from rocprofv3 --pmc passes
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if pmc.get("directed_edges") in (None, nE) and pmc.get("hidden") in (None, h) and args.mols == 512 and args.kind == "qm9":
                traffic = pmc.get("update_kernel_bytes_per_launch")
                if "roofline" in out and pmc.get("mega_kernel_bytes_per_launch"):
                    out["roofline"]["traffic"] = pmc.get("mega_kernel_bytes_per_launch")
                    out["roofline"]["traffic_source"] = ("profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the builder's run of "
                                                         "this command, scripts/pmc_traffic.py; NOT re-measured in this run)")
        except Exception:
            pass
        step_roof = {"kernel": kname, "bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_FP32_MFMA_TF,
                     "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TF, 4), "traffic": traffic,
                     "launch_us": round(t_k3 * 1e3, 3), "flop_per_launch": flops,
                     "algorithmic_bytes_per_launch": bytes_k3,
                     "algorithmic_GBps": round(bytes_k3 / (t_k3 * 1e-3) / 1e9, 1)}
        if "roofline" in out:
            out["roofline_per_step_kernel"] = step_roof   # the per-depth-step kernel of the fused route (large batches)
        else:
            out["roofline"] = step_roof
        # per-kernel breakdown of one forward (HIP events around each C-ABI call)
        try:
            Hn = torch.empty(nE, h, device=dev)
            Mv_ = torch.empty(nV, h, device=dev)
            kplan = lambda: engine.GraphPlan.from_bmg(bmg)
            run_steps(kplan, 5)
            br = {"plan_us": round(time_events(kplan, 30, torch) * 1e3, 2)}
            ktile = lambda: engine.GraphPlan.from_bmg(bmg, light="tiles")
            run_steps(ktile, 5)
            br["tile_plan_us"] = round(time_events(ktile, 30, torch) * 1e3, 2)  # what the inference forward builds (K0)
            if fusable:
                kagg = lambda: engine.update_fused(plan, Mbuf, H0buf, Wh, None, act="relu", want_M=False, want_Mv=True, Mv=Mv_)
                run_steps(kagg, 5)
                br["update_agg_us"] = round(time_events(kagg, 30, torch) * 1e3, 2)
            kfin = lambda: engine.linear(bmg.V, mp.W_o.weight.detach(), mp.W_o.bias.detach(), A2=Mv_, act="relu")
            run_steps(kfin, 5)
            br["finalize_us"] = round(time_events(kfin, 30, torch) * 1e3, 2)
            kini = lambda: engine.linear(bmg.V, mp.W_i.weight.detach(), None, A2=bmg.E, gather1=plan.src32, n_rows=nE, out=Hn)
            run_steps(kini, 5)
            br["init_unfused_us"] = round(time_events(kini, 30, torch) * 1e3, 2)
            out["kernel_breakdown"] = br
        except Exception as e:
            out["kernel_breakdown"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        # ---- the stand-alone scatter/gather step (K2, general route) against the HBM roofline ----
        Hbuf = torch.randn(nE, h, device=dev)
        Mout = torch.empty(nE, h, device=dev)
        k2 = lambda: engine.message(plan, Hbuf, out=Mout)
        run_steps(k2, 10)
        t_k2 = time_events(k2, 50, torch)
        bytes_k2 = 2.0 * nE * h * 4 + 3.0 * nE * 4
        gbs = bytes_k2 / (t_k2 * 1e-3) / 1e9
        out["roofline_scatter"] = {"kernel": "k_segment<message> (K2, general route)", "bound": "hbm", "achieved": round(gbs, 1),
                                   "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                                   "traffic": None, "launch_us": round(t_k2 * 1e3, 3), "bytes_per_launch": bytes_k2,
                                   "note": "working set fits the 256 MiB Infinity Cache at this batch"}
        try:
            if args.no_large_batches or world > 1:  # (multi-GPU runs: the other ranks wait at the final barrier meanwhile)
                raise RuntimeError("skipped (--no-large-batches)" if world == 1 else "skipped (N > 1: side measurements are taken at N = 1)")
            big = synth.random_batch(32768, args.kind, seed=5)
            big.to(dev)
            bplan = engine.GraphPlan.from_bmg(big)
            bE = int(big.E.shape[0])
            Hb = torch.randn(bE, h, device=dev)
            Mb = torch.empty(bE, h, device=dev)
            k2b = lambda: engine.message(bplan, Hb, out=Mb)
            run_steps(k2b, 3)
            t_b = time_events(k2b, 10, torch)
            bb = 2.0 * bE * h * 4 + 3.0 * bE * 4
            gb = bb / (t_b * 1e-3) / 1e9
            out["roofline_scatter_large"] = {"kernel": "k_segment<message> (K2), 32768 mols", "bound": "hbm",
                                             "achieved": round(gb, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                             "frac": round(gb / PEAK_HBM_GBS, 4), "traffic": None,
                                             "launch_us": round(t_b * 1e3, 3), "bytes_per_launch": bb,
                                             "directed_edges": bE}
            if fusable:  # the fused update at a batch whose working set exceeds the Infinity Cache
                H0b = torch.randn(bE, h, device=dev)
                Mn = torch.empty(bE, h, device=dev)
                k3b = lambda: engine.update_fused(bplan, Hb, H0b, Wh, None, act="relu", want_M=True, M_next=Mn)
                run_steps(k3b, 3)
                t3b = time_events(k3b, 10, torch)
                out["roofline_large"] = {"kernel": "fused per-depth update, 32768 mols", "bound": "mfma",
                                         "achieved": round(2.0 * bE * h * h / (t3b * 1e-3) / 1e12, 3),
                                         "peak": PEAK_FP32_MFMA_TF, "unit": "TFLOP/s",
                                         "frac": round(2.0 * bE * h * h / (t3b * 1e-3) / 1e12 / PEAK_FP32_MFMA_TF, 4),
                                         "launch_us": round(t3b * 1e3, 3), "directed_edges": bE,
                                         "algorithmic_GBps": round((3.0 * bE * h * 4 + 12.0 * bE) / (t3b * 1e-3) / 1e9, 1)}
                # the same per-step update as two kernels of the large-batch route: K2 (above) + the contraction on the
                # f16 pipe with the exact operand split (k_rows16), HBM bound: read M, read H0, write H
                Cb = torch.empty(bE, h, device=dev)
                wsb = {}
                k3c = lambda: engine.linear(Hb, Wh, None, Cadd=H0b, act="relu", out=Cb, mfma="split16")
                try:
                    run_steps(k3c, 3)
                    t3c = time_events(k3c, 10, torch)
                    b3 = 3.0 * bE * h * 4
                    out["roofline_steps16_large"] = {"kernel": "k_rows16<5> (+ k_split_weights): H = relu(H0 + M @ W_h^T), 3 x f16 MFMA on exactly split operands, 32768 mols",
                                                     "bound": "hbm", "achieved": round(b3 / (t3c * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS,
                                                     "unit": "GB/s", "frac": round(b3 / (t3c * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "traffic": None,
                                                     "launch_us": round(t3c * 1e3, 3), "bytes_per_launch": b3, "directed_edges": bE,
                                                     "TFLOPs": round(2.0 * bE * h * h / (t3c * 1e-3) / 1e12, 1)}
                except Exception as e:
                    out["roofline_steps16_large"] = {"error": f"{type(e).__name__}: {e}"[:200]}
                del H0b, Mn, Cb
            del big, bplan, Hb, Mb
        except Exception as e:
            out["roofline_scatter_large"] = {"error": f"{type(e).__name__}: {e}"[:200]}

        # ---- host hand-off (f3): PCIe-inclusive rates — never `value`.  (a) the reference's hand-off: five host tensors
        # (int64 indices) moved one by one (collate.py:68-73); (b) one pinned buffer + dmpnn_collate.  Wall clock with a
        # device synchronize on both sides; packing the buffer is the DataLoader worker's job and is not in (b). ----
        if world == 1 and not train:
            try:
                from chemprop_amd.data import BatchMolGraph, PackedBatch

                mgs = synth.random_molgraphs(args.mols, args.kind, seed=1000 + rank)
                host_bmg = BatchMolGraph(mgs)
                for k in ("V", "E", "edge_index", "rev_edge_index", "batch"):
                    setattr(host_bmg, k, getattr(host_bmg, k).pin_memory())
                packed = PackedBatch(mgs, pin=True)

                def wall(fn, n=30):
                    for _ in range(5):
                        fn()
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    for _ in range(n):
                        fn()
                    torch.cuda.synchronize(dev)
                    return (time.perf_counter() - t0) / n * 1e6

                def five():
                    b = BatchMolGraph.from_tensors(*(getattr(host_bmg, k).to(dev, non_blocking=True)
                                                     for k in ("V", "E", "edge_index", "rev_edge_index", "batch")), len(mgs))
                    with torch.no_grad():
                        return mp(b)

                def one():
                    with torch.no_grad():
                        return mp(packed.to_device(dev))

                t5, t1 = wall(five), wall(one)
                t_c = wall(lambda: packed.to_device(dev))
                out["host_handoff"] = {
                    "five_tensors_us": round(t5, 1), "packed_us": round(t1, 1), "packed_copy_and_collate_us": round(t_c, 1),
                    "wire_bytes": int(packed.buf.numel()),
                    "five_tensor_bytes": int(sum(getattr(host_bmg, k).numel() * getattr(host_bmg, k).element_size()
                                                 for k in ("V", "E", "edge_index", "rev_edge_index", "batch"))),
                    "pcie_inclusive_M_edge_updates_per_s": round(updates / t1, 2),
                    "pcie_inclusive_five_tensors_M_edge_updates_per_s": round(updates / t5, 2),
                    "note": "pinned host buffers -> device -> forward, eager, wall clock; weights resident"}
                # the same forward on a RESIDENT batch that came through the packed format: its tile table was made by the
                # loader (dmpnn_pack_tiles), so K0 is a copy of that table and the tile kernel is not limited to small batches
                lt = {}
                for n_m in sorted({args.mols} if args.no_large_batches else {args.mols, 4096}):
                    mg2 = synth.random_molgraphs(n_m, args.kind, seed=1000 + rank)
                    pb = PackedBatch(mg2, pin=True)
                    if pb.n_tiles <= 0:
                        lt[str(n_m)] = {"note": "a molecule exceeds a tile: no loader table"}
                        continue
                    res = pb.to_device(dev)
                    plain = BatchMolGraph(mg2)
                    plain.to(dev)
                    upd = int(res.E.shape[0]) * (args.depth - 1)
                    ent = {"directed_edges": int(res.E.shape[0]), "tiles": pb.n_tiles}
                    for tag, b in (("loader_tiles", res), ("device_plan", plain)):
                        def f(b=b):
                            with torch.no_grad():
                                return mp(b)
                        run_steps(f, 10)
                        t = time_events(f, 50, torch)
                        ent[tag + "_us"] = round(t * 1e3, 2)
                        ent[tag + "_M_edge_updates_per_s"] = round(upd / (t * 1e3), 2)
                    lt[str(n_m)] = ent
                out["loader_tiles"] = lt
            except Exception as e:
                out["host_handoff"] = {"error": f"{type(e).__name__}: {e}"[:200]}

        # ---- BASELINE configs[1] names bf16: the SAME forward with every matrix product on f16 operands (DMPNN_STORE=f16 on the tile
        # route, round 6: operands, messages and weights as one f16 per element under the exact split's power-of-two scales, one MFMA
        # pass instead of three; 11-bit significands, bf16 has 8).  OPT-IN, NOT fp32-class: reported beside `value`, never as it ----
        if world == 1 and not train:
            try:
                def f_lp():
                    with torch.no_grad():
                        return mp(bmg)
                out_exact = f_lp().clone()
                os.environ["DMPNN_STORE"] = "f16"
                try:
                    run_steps(f_lp, 10)
                    t_lp = time_events(f_lp, max(50, args.steps), torch)
                    o_lp = f_lp()
                    den = max(1.0, float(out_exact.abs().max()))
                    out["half_operands"] = {"us": round(t_lp * 1e3, 2), "M_edge_updates_per_s": round(updates / (t_lp * 1e3), 2),
                                            "route": mp.__dict__.get("_dmpnn_route"),
                                            "max_diff_vs_exact_over_max_out": float((o_lp - out_exact).abs().max()) / den,
                                            "note": "DMPNN_STORE=f16: opt-in, not fp32-class (tests hold 2e-3 against the oracle and <= the oracle under "
                                                    "torch's bf16 autocast); never `value`"}
                finally:
                    os.environ["DMPNN_STORE"] = "f32"
                    run_steps(f_lp, 3)
            except Exception as e:
                out["half_operands"] = {"error": f"{type(e).__name__}: {e}"[:200]}

        # ---- BASELINE configs[2..4] at their own shapes (inference forward of the module, inputs resident): molecules that
        # do not fit the tiles of the whole-forward kernel take the per-step fused route on the f16 pipe ----
        if world == 1 and not train and not args.no_large_batches:
            oc = {}
            for name, kind, n_m, kw in (("zinc-512 h512 d6 (configs[2])", "zinc", 512, dict(d_h=512, depth=6)),
                                        ("synth40-512 (configs[3], 512 mols/GPU)", "synth40", 512, dict()),
                                        ("synth40-4096 (configs[3], 4096 mols/GPU)", "synth40", 4096, dict()),
                                        ("cgr-512 (configs[4])", "cgr", 512, dict(d_v=106, d_e=28)),
                                        ("qm9-4096 (the headline shape at 4096 mols/GPU)", "qm9", 4096, dict()),
                                        ("qm9-512 h1200 (hpopt's width range, cli/hpopt.py:73)", "qm9", 512, dict(d_h=1200))):
                try:
                    b2 = synth.random_batch(n_m, kind, seed=1)
                    b2.to(dev)
                    torch.manual_seed(0)
                    m2 = BondMessagePassing(**kw).eval().to(dev)

                    def f2():
                        with torch.no_grad():
                            return m2(b2)
                    run_steps(f2, 5)
                    t2 = time_events(f2, 20, torch)
                    e2 = int(b2.E.shape[0])
                    oc[name] = {"directed_edges": e2, "us": round(t2 * 1e3, 1), "M_edge_updates_per_s": round(e2 * (m2.depth - 1) / (t2 * 1e3), 1),
                                "route": m2.__dict__.get("_dmpnn_route")}
                    # its own roofline block (round-4 VERDICT item 2): SURVEY 8(d)'s forward flops against the f16 pipe's dense peak / 3
                    # passes of the exact split, and — for the per-step routes, whose messages travel through HBM — the algorithmic bytes
                    # of the whole forward (per depth step: read M, write M_next as split rows + the residual operand; K1 / finalize
                    # operands; indices) against 8 TB/s.  Times are the WHOLE forward incl. K0 (the routes' kernels are not separated here:
                    # profiles/ holds their kernel stats)
                    v2, h2, dv2, de2, dp2 = int(b2.V.shape[0]), int(m2.W_h.weight.shape[0]), int(b2.V.shape[1]), int(b2.E.shape[1]), int(m2.depth)
                    fl2 = 2.0 * e2 * (dv2 + de2) * h2 + 2.0 * e2 * h2 * h2 * (dp2 - 1) + 2.0 * v2 * (dv2 + h2) * h2
                    row = 4 * h2 + 16
                    by2 = (4.0 * (v2 * dv2 + e2 * de2) + 4.0 * v2 * h2 + 12.0 * e2 if str(oc[name]["route"]).startswith("mega") else
                           4.0 * (v2 * dv2 + e2 * de2) + e2 * (4.0 * (dv2 + de2) + 16) * (dp2 if h2 <= 320 else 1) + (0 if h2 <= 320 else 4.0 * e2 * h2 * dp2)
                           + 2.0 * e2 * row * (dp2 - 1) + 2.0 * v2 * row + 4.0 * v2 * (dv2 + h2) + 12.0 * e2 * dp2)
                    oc[name]["roofline"] = {"flop": fl2, "TFLOP_per_s": round(fl2 / (t2 * 1e-3) / 1e12, 1), "peak_TFLOP_per_s": 833.3,
                                            "frac_mfma": round(fl2 / (t2 * 1e-3) / 1e12 / 833.3, 4), "algorithmic_bytes": by2,
                                            "GB_per_s": round(by2 / (t2 * 1e-3) / 1e9, 1), "frac_hbm": round(by2 / (t2 * 1e-3) / 8e12, 4),
                                            "bound": "mfma" if str(oc[name]["route"]).startswith("mega") else "simd issue (MFMA + VALU epilogue), see DESIGN section 4"}
                    # the same forward with the OPT-IN half storage of the messages (DMPNN_F_STORE16: NOT fp32-class, ~1e-4
                    # relative; reported beside the exact figure, never instead of it)
                    os.environ["DMPNN_STORE"] = "f16"
                    try:
                        run_steps(f2, 5)
                        t3 = time_events(f2, 20, torch)
                        oc[name]["f16_storage_us"] = round(t3 * 1e3, 1)
                        oc[name]["f16_storage_route"] = m2.__dict__.get("_dmpnn_route")
                    finally:
                        os.environ["DMPNN_STORE"] = "f32"
                    if kind in ("synth40", "qm9"):
                        # BASELINE configs[3] is a TRAINING workload: forward (kept tensors) + backward + fused Adam of this shape
                        # (qm9-4096: beyond the single-workgroup plan — the tile kernels on a full plan with molecule tiles)
                        from chemprop_amd import distributed as ddp2
                        from chemprop_amd.optim import FlatAdam as FlatAdam2

                        m3 = BondMessagePassing(**kw).to(dev).train()
                        s3 = ddp2.GradSync(list(m3.parameters()), modules=[m3])
                        o3 = FlatAdam2(s3, lr=1e-4)
                        G3 = torch.randn(int(b2.V.shape[0]), m3.output_dim, device=dev)

                        def f3():
                            with ddp2.backward_on_calling_thread():
                                m3(b2).backward(G3)
                            s3.allreduce()
                            o3.step()
                        run_steps(f3, 3)
                        t4 = time_events(f3, 10, torch)
                        oc[name]["train_step_us"] = round(t4 * 1e3, 1)
                        oc[name]["train_M_edge_updates_per_s"] = round(e2 * (m3.depth - 1) / (t4 * 1e3), 1)
                        try:
                            st3 = m3(b2).grad_fn.st                             # (the forward state of the autograd node)
                            oc[name]["train_route"] = st3.route
                            oc[name]["train_operands"] = "split rows (k_wgrad16r)" if (st3.args.msplit or st3.route == "fused16/lean") else "blocks (k_wsplit16 + k_wgrad16)"
                        except AttributeError:
                            oc[name]["train_route"] = None
                        del m3, s3, o3, G3
                    del b2, m2
                except Exception as e:
                    oc[name] = {"error": f"{type(e).__name__}: {e}"[:200]}
            try:  # f2: AtomMessagePassing inference at the headline shape (the tile kernel with DMPNN_F_ATOM) beside the bond block
                from chemprop_amd.nn import AtomMessagePassing

                torch.manual_seed(0)
                am = AtomMessagePassing(d_v=d_v, d_e=d_e, d_h=args.hidden, depth=args.depth).eval().to(dev)

                def fa():
                    with torch.no_grad():
                        return am(bmg)
                run_steps(fa, 6)
                ta = time_events(fa, 50, torch)
                oc[f"atom-{args.kind}-{args.mols} (AtomMessagePassing, inference)"] = {
                    "directed_edges": nE, "us": round(ta * 1e3, 1), "M_edge_updates_per_s": round(updates / (ta * 1e3), 1),
                    "route": am.__dict__.get("_dmpnn_route"), "bond_block_us": round(ms_per_step * 1e3, 1)}
                del am
                # ... and its TRAINING step on the tile kernels (round 4: DMPNN_F_ATOM | DMPNN_F_KEEP + the backward tile kernel), measured
                # like `train_step` of the bond block: forward (kept) + backward + gradient exchange + flat Adam
                from chemprop_amd import distributed as ddp3
                from chemprop_amd.optim import FlatAdam as FlatAdam3

                at = AtomMessagePassing(d_v=d_v, d_e=d_e, d_h=args.hidden, depth=args.depth).to(dev).train()
                s4 = ddp3.GradSync(list(at.parameters()), modules=[at])
                o4 = FlatAdam3(s4, lr=1e-4)
                G4 = torch.randn(nV, args.hidden, device=dev)

                def ft():
                    with ddp3.backward_on_calling_thread():
                        at(bmg).backward(G4)
                    s4.allreduce()
                    o4.step()
                run_steps(ft, 6)
                tt = time_events(ft, 30, torch)
                oc[f"atom-{args.kind}-{args.mols} (AtomMessagePassing, inference)"].update(
                    {"train_step_us": round(tt * 1e3, 1), "train_route": at.__dict__.get("_dmpnn_route"),
                     "bond_block_train_step_us": (round(out["train_step"]["ms_per_step"] * 1e3, 1) if "ms_per_step" in out.get("train_step", {}) else None)})
                del at, s4, o4, G4
            except Exception as e:
                oc["atom"] = {"error": f"{type(e).__name__}: {e}"[:200]}
            out["other_configs"] = oc

        # ---- CPU baseline: the EXECUTED reference (chemprop.nn.BondMessagePassing from oracle/_ref or /root/reference under the
        # import shim) on the host cores, bounded sample; the restated op sequence (oracle/dmpnn_torch.py) only where no
        # reference tree travelled with the snapshot.  Forward (beside `value`) and forward + backward (beside `train_step`). ----
        if world == 1 and not args.no_cpu_baseline:
            kind_cpu, cpu_fwd, cpu_train, what = None, None, None, None
            try:
                from oracle import ref_shim

                if ref_shim.reference_available():
                    BMP, BMG, _MG = ref_shim.load_reference()
                    ref_mp = BMP(d_v=d_v, d_e=d_e, d_h=args.hidden, depth=args.depth)
                    ref_mp.load_state_dict(cpu_state)
                    ref_bmg = BMG(synth.random_molgraphs(args.mols, args.kind, seed=1000 + rank))

                    def cpu_fwd():
                        ref_mp.eval()
                        with torch.no_grad():
                            ref_mp(ref_bmg)

                    def cpu_train():  # out.sum().backward() through the reference's own ATen ops
                        ref_mp.train()
                        for p_ in ref_mp.parameters():
                            p_.grad = None
                        ref_mp(ref_bmg).sum().backward()
                    kind_cpu = "reference"
                    what = (f"the executed reference: chemprop.nn.BondMessagePassing.forward from {ref_shim.REFERENCE_ROOT} under "
                            "oracle/ref_shim.py (its own source, torch CPU kernels), on the reference's own BatchMolGraph")
            except Exception as e:  # report and fall back to the port
                out["cpu_baseline_reference_error"] = f"{type(e).__name__}: {e}"[:200]
            if kind_cpu is None:
                from oracle import dmpnn_torch as ot

                cpu_mp = BondMessagePassing(d_v=d_v, d_e=d_e, d_h=args.hidden, depth=args.depth)
                cpu_mp.load_state_dict(cpu_state)
                w = ot.MPWeights.from_module(cpu_mp)

                def cpu_fwd():
                    with torch.no_grad():
                        ot.forward_bmg(cpu_bmg, w, depth=args.depth)

                def cpu_train():
                    for t in (w.W_i, w.W_h, w.W_o, w.b_o):
                        t.requires_grad_(True)
                        t.grad = None
                    ot.forward_bmg(cpu_bmg, w, depth=args.depth).sum().backward()
                kind_cpu = "port"
                what = ("oracle/dmpnn_torch.py (the reference's ATen op sequence, checked against the executed reference on the "
                        "goldens): no reference tree (oracle/_ref) travelled with this snapshot")

            # thread sweep: torch's default (every hardware thread: 128 on the GPU box) oversubscribes a 9 k-edge problem by
            # 10x; the baseline is the BEST median over {1, 8, 16, 32, all} threads, each timed for an equal share of the
            # budget, and the thread count that won is what `cores` reports
            all_threads = torch.get_num_threads()
            sweep = sorted({t for t in (1, 8, 16, 32, all_threads) if t <= all_threads})

            def cpu_sweep(fn, seconds):
                per = {}
                for nt in sweep:
                    torch.set_num_threads(nt)
                    fn(); fn()
                    times = []
                    t_end = time.perf_counter() + seconds / len(sweep)
                    while time.perf_counter() < t_end or len(times) < 3:
                        t0 = time.perf_counter()
                        fn()
                        times.append(time.perf_counter() - t0)
                    times.sort()
                    per[nt] = (times[len(times) // 2], times[0], len(times))
                torch.set_num_threads(all_threads)
                cores = min(per, key=lambda k: per[k][0])
                return cores, per

            def cpu_entry(fn, seconds, leg):
                cores, per = cpu_sweep(fn, seconds)
                med, best, n_rep = per[cores]
                return {"value": round(updates / med / 1e6, 4), "unit": "M edge-updates/s", "cores": cores, "kind": kind_cpu,
                        "sample": f"{leg}: {n_rep} repetitions (~{seconds / len(sweep):.0f} s per thread count) of the same {args.mols}-molecule "
                                  f"batch; {what}; torch {torch.__version__} CPU; median at the best of {sweep} threads ({cores}); "
                                  f"host has {all_threads} hardware threads",
                        "ms_per_step": round(med * 1e3, 3), "min_ms_per_step": round(best * 1e3, 3),
                        "median_ms_by_threads": {str(k): round(v[0] * 1e3, 3) for k, v in per.items()}}

            main_fn, other_fn = (cpu_train, cpu_fwd) if train else (cpu_fwd, cpu_train)
            out["cpu_baseline"] = cpu_entry(main_fn, args.cpu_seconds * 0.6, "forward + backward" if train else "forward (eval, no_grad)")
            out["speedup_vs_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
            if not train:
                out["cpu_baseline_train"] = cpu_entry(other_fn, args.cpu_seconds * 0.4, "forward + backward (out.sum().backward())")
                ts = out.get("train_step", {})
                if "M_edge_updates_per_s" in ts:
                    out["train_speedup_vs_cpu"] = round(ts["M_edge_updates_per_s"] / out["cpu_baseline_train"]["value"], 1)
        # Two lines: everything (prose notes, per-thread CPU sweeps, every side measurement's bookkeeping) first, tagged; then the
        # LAST line — the one the driver parses and whose tail it records — numbers only, a few KB, the training / whole-model /
        # other-config figures at its end (round-5 VERDICT weak #7: a 14 KB line pushed model_step out of the recorded tail)
        print("BENCH_DETAIL " + json.dumps(out), flush=True)
        print(json.dumps(compact_line(out)), flush=True)
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
