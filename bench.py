#!/usr/bin/env python3
"""Benchmark of the hot path: one full training step (forward + losses + backward + gradient all-reduce + fused
Adam) on synthetic part clouds, B = 32 per GPU, P = 20, N = 1000 (weak scaling over GPUs).

    python bench.py [--gpus N --steps K --warmup W] [--config c1|c2|c3|c4|c5 | global_partnet|pn_transformer|dgl_dgcnn|rgl_net_dgcnn]

With --gpus N > 1 and no torch.distributed environment the script re-launches itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (one rank per GPU over
RCCL); launched by the driver under torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE from the env.

Workloads (BASELINE.json configs[0..4]):
    c2 (default) pn_transformer + PointNet, Breaking-Bad-everyday-like clouds       — the configuration `metric` is quoted on
    c4           = c2 per GPU (the 8-GPU line is `--gpus 8` of the same workload)
    c3           DGL (3 GNN iterations) + DGCNN encoder, everyday-like clouds        — kNN / EdgeConv kernel stress
    c5           RGL-NET + DGCNN, artifact-like clouds (12-20 small parts per shape)
    c1           B-Global + PointNet, semantic flags on (matching, min-of-5), P = 2, B = 4 — plumbing case

Rank 0 prints ONE JSON line.  `value` = parts (B x P slots, padded slots included, as the metric is defined)
processed per second by the whole job, inputs resident in HBM before the timed region.  Kernels are launched
eagerly (`--graph` replays the captured step as one HIP graph).  The garbage collector is parked during the timed
steps (a generation-2 collection inside K steps of 3 ms was measured as +1.6 ms per step).

`roofline` describes the dominant kernel of the workload — c1/c2/c4: the whole-shape Chamfer search of the fused
loss; c3/c5: the kNN graph kernel of the DGCNN encoder — timed per launch with HIP events that the library records
around it on its launch stream inside the timed region.  It carries the mandated HBM comparison (algorithmic bytes of
SURVEY.md §8d per launch / time / 8 TB/s) AND the bound that actually binds (`binding`), plus a per-kernel table
(`kernel_table`) with MFMA / VALU / HBM fractions of the other large kernels.  `cpu_baseline` is the oracle's
reference-equivalent PyTorch-CPU step of the SAME workload timed on this host (rank 0, N = 1) on a bounded sample: c1 at
full size, c2 / c4 at B = 4, c3 / c5 at B = 2 (scaled linearly in B, labelled), on all physical cores and on one thread.
With several ranks the line carries `collectives`: bytes of each gradient bucket, its stand-alone all-reduce time, the step
time without any collective, and the fraction of the collective time that backward hid (`overlap_frac`).
`--self-check N` appends N more timed steps and reports whether the K-step mean holds over them.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

BATCH, PARTS, POINTS = 32, 20, 1000
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_LANE_OPS = 78.6e12   # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz (non-packed fp32 VALU issue)
MFMA_F32_PEAK = 157.3e12       # 256 CU x 4 SIMD x 64 FLOP/cycle x 2.4 GHz (v_mfma_f32_32x32x2_f32, dense)
MFMA_BF16_PEAK = 2.5e15        # dense bf16 (MI355X_MICROARCH.md; 1.75-1.9e15 sustained: tools/probes/mfma_rate.hip)


CONFIG_ALIASES = {"global_partnet": "c1", "pn_transformer": "c2", "dgl_dgcnn": "c3", "pn_transformer_dp8": "c4",
                  "rgl_net_dgcnn": "c5"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 20 untimed + 100 timed steps (0.3 s for c2, 3 s for c5).  The mean of the first 20 steps after 5 warm-up steps
    # sits ~3 % above the steady state (2.44 vs 2.37 ms for c2; 300 steps: 2.35) - `--self-check N` reports that ratio.
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=["c1", "c2", "c3", "c4", "c5"] + sorted(CONFIG_ALIASES))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batches", type=int, default=N_BATCHES,
                    help="distinct synthetic batches rotated through the steps (1: one static batch, as rounds 1-3 did)")
    ap.add_argument("--no-chamfer-standalone", action="store_true",
                    help="skip the stand-alone timing of the drop-in Chamfer operator (the `chamfer_standalone` object)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as one HIP graph (default: eager launches; see trainer.py)")
    ap.add_argument("--eager", action="store_true", help="(the default; accepted for symmetry)")
    ap.add_argument("--cpu-batch", type=int, default=0,
                    help="samples in the CPU-baseline step (0: the workload's default: c1 4 = full size, c2/c4 4, c3 2)")
    ap.add_argument("--self-check", type=int, default=0, metavar="N",
                    help="after the K timed steps, time N more (same protocol) and report the ratio of the two means")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="bf16: the separately named PERFORMANCE VARIANT of the PointNet encoder (csrc/pointnet_bf16.hip: "
                         "bf16 activations and matrix-core GEMMs, fp32 statistics; transformer, pose head, losses and "
                         "optimiser stay fp32) - a second line beside the fp32 parity run, never the default")
    args = ap.parse_args()
    args.config = CONFIG_ALIASES.get(args.config, args.config)
    return args


def dp_backend():
    """`MPA_DP_BACKEND=gloo` runs the N > 1 path (process group, bucketed all-reduce under backward, the `collectives`
    object) over gloo with the ranks dealt round-robin onto the visible GPUs: every line of the data-parallel branch
    executes on a 1-GPU box (tests/test_zz_bench_gpu.py).  Default `nccl` (= RCCL over xGMI), one rank per GPU."""
    b = os.environ.get("MPA_DP_BACKEND", "nccl").lower()
    if b not in ("nccl", "gloo"):
        sys.exit(f"bench.py: MPA_DP_BACKEND={b!r} (nccl or gloo)")
    return b


def relaunch_distributed(args):
    """`python bench.py --gpus N` with N > 1 outside torch.distributed.run: start one rank per GPU ourselves."""
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus and dp_backend() == "nccl":  # (gloo: ranks may share a GPU — the N > 1 code path without N GPUs)
        sys.exit(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    sys.exit(subprocess.run(cmd, env=env).returncode)


# ---- workloads ----------------------------------------------------------------------------------------------------
N_BATCHES = 4  # distinct synthetic batches rotated through the warm-up and timed steps (batch k: seed 1234 + rank + 1000 k);
               # `--batches 1` replays batch 0 alone, the protocol of rounds 1-3 (353 valid parts)


def representative_parts(preset, seed, batch=BATCH, max_parts=PARTS, tol=4):
    """Part counts of one batch, drawn like synthetic.make_batch draws them (uniform in the preset's range) but REDRAWN until
    their sum is within `tol` of its expectation (everyday: 32 x 11 = 352 valid parts, artifact: 32 x 16 = 512).  The metric
    counts B x P slots per step while the work follows the VALID parts; a handful of random batches is a noisy sample of the
    long-run mix (the first four seeds drew 353 / 329 / 394 / 414), and across ranks the heaviest draw sets the step time.
    Batches near the expectation measure the long-run throughput, and every rank gets the same amount of work."""
    import torch
    from multi_part_assembly_amd.synthetic import PRESETS
    lo, hi = PRESETS[preset]["min_parts"], min(PRESETS[preset]["max_parts"], max_parts)
    want = batch * (lo + hi) / 2.0
    for attempt in range(10000):
        g = torch.Generator(device="cpu").manual_seed(seed + 7919 * attempt)
        parts = torch.randint(lo, hi + 1, (batch,), generator=g).tolist()
        if abs(sum(parts) - want) <= tol:
            return parts
    raise RuntimeError("no representative draw")


def workload(name, rank, dev, k=0):
    """-> (cfg, batch k of this rank, description, B, P).  Batch 0 of rank 0 is the batch of rounds 1-3 (seed 1234: 353 valid
    parts, 526 for the artifact preset); every other batch has a representative part count (representative_parts)."""
    from multi_part_assembly_amd import config, synthetic
    historical = rank == 0 and k == 0
    rank = rank + 1000 * k
    parts_of = lambda preset: None if historical else representative_parts(preset, 1234 + rank)
    if name in ("c2", "c4"):
        cfg = config.pn_transformer_everyday()
        batch = synthetic.make_batch(BATCH, PARTS, POINTS, preset="everyday", seed=1234 + rank, device=dev,
                                     num_parts=parts_of("everyday"))
        desc = ("pn_transformer + PointNet encoder, Breaking-Bad-everyday-like synthetic clouds, B=32 per GPU, P=20, "
                "N=1000, geometric loss, Adam (BASELINE.json configs[1]; configs[3] = the same workload on 8 GPUs)")
        return cfg, batch, desc, BATCH, PARTS
    if name == "c3":
        cfg = config.dgl_dgcnn_everyday()
        batch = synthetic.make_batch(BATCH, PARTS, POINTS, preset="everyday", seed=1234 + rank, device=dev,
                                     num_parts=parts_of("everyday"))
        desc = ("DGL (3 GNN iterations, merge_node off) + DGCNN encoder, everyday-like synthetic clouds, B=32 per "
                "GPU, P=20, N=1000, geometric loss summed over the iterations, Adam (BASELINE.json configs[2])")
        return cfg, batch, desc, BATCH, PARTS
    if name == "c5":
        cfg = config.rgl_net_dgcnn_artifact()
        batch = synthetic.make_batch(BATCH, PARTS, POINTS, preset="artifact", seed=1234 + rank, device=dev,
                                     num_parts=parts_of("artifact"))
        desc = ("RGL-NET (bi-GRU, 3 iterations) + DGCNN encoder, artifact-like synthetic clouds (12-20 small parts), "
                "B=32 per GPU, P=20, N=1000, Chamfer in fp32 as the reference forces it (BASELINE.json configs[4])")
        return cfg, batch, desc, BATCH, PARTS
    cfg = config.global_partnet_chair()
    cfg.data.max_num_part = 2
    batch = synthetic.make_semantic_batch(4, 2, POINTS, seed=1234 + rank, device=dev,
                                          num_part_category=cfg.data.num_part_category)
    desc = ("B-Global + PointNet, semantic flags on (identical-part matching, min-of-5 sampling, 32 noise channels), "
            "P=2, B=4, N=1000 (BASELINE.json configs[0]: the reference's CPU-runnable plumbing case)")
    return cfg, batch, desc, 4, 2


# ---- CPU baseline ---------------------------------------------------------------------------------------------------
def cpu_model_name():
    try:
        for line in Path("/proc/cpuinfo").read_text().splitlines():
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores():
    """Distinct (package, core) pairs among the CPUs this process may run on; falls back to the visible count."""
    allowed = os.sched_getaffinity(0)
    try:
        seen, cur = set(), {}
        for line in Path("/proc/cpuinfo").read_text().splitlines() + [""]:
            if not line.strip():
                if "processor" in cur and int(cur["processor"]) in allowed:
                    seen.add((cur.get("physical id", "0"), cur.get("core id", cur["processor"])))
                cur = {}
            elif ":" in line:
                k, v = line.split(":", 1)
                cur[k.strip()] = v.strip()
        if seen:
            return len(seen)
    except (OSError, ValueError):
        pass
    return len(allowed)


def cpu_baseline(name, cpu_batch, dev):
    """Reference-equivalent CPU training step (oracle/nets.py, oracle/callers.py + oracle/chamfer_ref.c) on a bounded
    sample of the workload `name`: the SAME synthetic generator and seed as the GPU run at B = cpu_batch (cost is linear
    in B; c1 runs at its full size), on all physical cores and on one thread, forward + losses + backward + Adam."""
    import torch
    from multi_part_assembly_amd import config, synthetic
    from multi_part_assembly_amd.pn_transformer import build_model
    from oracle import callers as oc
    from oracle import nets as on

    visible = len(os.sched_getaffinity(0))
    cores = physical_cores()
    P, N = PARTS, POINTS
    if name in ("c2", "c4"):
        cfg, full_B, B_all, B_one, label = config.pn_transformer_everyday(), BATCH, cpu_batch or 4, 2, "c2"
        make = lambda B: synthetic.make_batch(B, P, N, preset="everyday", seed=1234, device=dev)
    elif name == "c3":
        cfg, full_B, B_all, B_one, label = config.dgl_dgcnn_everyday(), BATCH, cpu_batch or 2, 1, "c3"
        make = lambda B: synthetic.make_batch(B, P, N, preset="everyday", seed=1234, device=dev)
    elif name == "c5":
        cfg, full_B, B_all, B_one, label = config.rgl_net_dgcnn_artifact(), BATCH, cpu_batch or 2, 1, "c5"
        make = lambda B: synthetic.make_batch(B, P, N, preset="artifact", seed=1234, device=dev)
    elif name == "c1":
        cfg = config.global_partnet_chair()
        cfg.data.max_num_part = 2
        P, full_B, B_all, B_one, label = 2, 4, cpu_batch or 4, 4, "c1"
        make = lambda B: synthetic.make_semantic_batch(B, 2, N, seed=1234, device=dev,
                                                       num_part_category=cfg.data.num_part_category)
    else:
        return None
    torch.manual_seed(0)
    model = build_model(cfg)
    loss_cfg = {k: cfg.loss[k] for k in cfg.loss}

    def run(B, threads, warm, timed):
        torch.set_num_threads(threads)
        os.environ["OMP_NUM_THREADS"] = str(threads)
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        params = {k: sd[k].requires_grad_() for k, _ in model.named_parameters()}
        batch = {k: v.cpu() for k, v in make(B).items() if hasattr(v, "cpu")}
        state = {}

        def step():
            for p in params.values():
                p.grad = None
            stats = {}
            if label == "c2":
                losses, _ = on.pn_transformer_loss(sd, batch, cfg.model.transformer_layers,
                                                   cfg.model.transformer_heads, training=True, stats_out=stats)
            elif label in ("c3", "c5"):
                losses = oc.dgl_loss(sd, batch, cfg.model.gnn_iter, cfg.model.encoder, True, stats, recurrent=label == "c5",
                                     merge_node=cfg.model.merge_node)
            else:
                losses = oc.global_loss(sd, batch, loss_cfg, cfg.loss.sample_iter, cfg.loss.noise_dim, cfg.model.encoder,
                                        True, stats)
            losses["loss"].backward()
            live = {k: p for k, p in params.items() if p.grad is not None}
            on.adam_step(live, {k: p.grad for k, p in live.items()}, state, lr=cfg.optimizer.lr)
            for k, v in stats.items():
                sd[k] = v

        for _ in range(warm):
            step()
        t0 = time.perf_counter()
        for _ in range(timed):
            step()
        dt = (time.perf_counter() - t0) / timed
        return B * P / dt, dt

    heavy = label in ("c3", "c5")  # ~8 s per step: 1 warm-up + 1 timed per thread count keeps the default run within minutes
    # torch's CPU ops do not scale to 128 threads at these sizes (c2: 57 parts/s on 128 threads, 100 on 64): the baseline
    # is the BEST of {all physical cores, 64, 16} threads, and every measured point is reported
    by_threads = {}
    for threads in sorted({cores, min(64, cores), min(16, cores)}, reverse=True):
        by_threads[threads] = run(B_all, threads, 1 if heavy else 2, 1 if heavy else 4)
    best = max(by_threads, key=lambda t: by_threads[t][0])
    v_all, dt_all = by_threads[best]
    v_one, dt_one = run(B_one, 1, 0 if heavy else 1, 1 if heavy else 2)
    scaled = "" if B_all == full_B else f" (vs {full_B}: cost linear in B)"
    return {"value": v_all, "unit": "parts/s", "cores": best, "kind": "port", "physical_cores": cores,
            "by_threads": {str(t): v for t, (v, _) in by_threads.items()},
            "one_thread_value": v_one, "cpu": cpu_model_name(), "visible_cores": visible,
            "sample": f"full train step (fwd+loss+bwd+Adam) of the {label} model on the bench's own generator (seed 1234) "
                      f"at B={B_all}{scaled}, P={P}, N={N}: {dt_all:.2f} s/step on {best} threads, the best of "
                      f"{sorted(by_threads, reverse=True)} ({cores} physical cores, {visible} hardware threads visible); "
                      f"one thread: B={B_one}, {dt_one:.2f} s/step; torch CPU ops + OpenMP C Chamfer (oracle/)"}


# ---- roofline bookkeeping -------------------------------------------------------------------------------------------
def _find(kernels, prefix):
    return [(k, v) for k, v in kernels.items() if k.startswith(prefix)]


def kernel_table(kernels, nv, cfg, B, P):
    """Per-entry-point fractions of the bound that binds each of the large kernels (algorithmic work / measured time); nv =
    valid parts per step (mean over the rotated batches).  Every row names its peak (`peak`) and what `frac` is a fraction
    of; matrix-core rows count the instructions the kernels actually issue (the fp32-grade split-bf16 products cost SIX bf16
    matrix instructions per fp32 product, so their `frac` is matrix-pipe occupancy at the bf16 rate, and `fp32_equiv_tflops`
    is the useful rate)."""
    N = POINTS
    F = cfg.model.pc_feat_dim
    rows = {}

    def per_step(prefix):
        hit = _find(kernels, prefix)
        if not hit:
            return None
        return sum(v["total_ms"] for _, v in hit) / max(1, hit[0][1]["launches"])

    def add_mfma(prefix, name, flops_f32=0.0, flops_split=0.0, useful=None, note=None):
        """flops_f32: FLOP issued as v_mfma_f32_32x32x2_f32; flops_split: fp32-grade FLOP issued as 6 bf16 products."""
        ms = per_step(prefix)
        if ms is None:
            return
        secs = ms * 1e-3
        pipe_s = flops_f32 / MFMA_F32_PEAK + 6.0 * flops_split / MFMA_BF16_PEAK  # matrix-pipe seconds at the two peaks
        useful = useful if useful is not None else flops_f32 + flops_split
        rows[name] = {"ms_per_step": ms, "bound": "matrix-pipe", "frac": pipe_s / secs,
                      "peak": "time the issued matrix instructions need at their peak rates (v_mfma_f32_32x32x2_f32 157.3 "
                              "TFLOP/s; v_mfma_f32_32x32x16_bf16 2.5 PFLOP/s, six products per fp32-grade product) / "
                              "measured time = matrix-pipe occupancy",
                      "flops_exact_f32_mfma": flops_f32, "flops_fp32_grade_split_bf16": flops_split,
                      "bf16_products_per_fp32_product": 6, "fp32_equiv_tflops": useful / secs / 1e12,
                      **({"note": note} if note else {})}

    if cfg.model.encoder == "pointnet":
        r = nv * N
        # forward: conv1 on the VALU (negligible), conv2..conv5 split-bf16 (csrc/pn_fwd_ws.h)
        add_mfma("pointnet_forward[", "pointnet_forward", flops_split=2.0 * r * (64 * 64 + 64 * 64 + 64 * 128 + 128 * F))
        # backward in Q form (csrc/pn_bwd_q.h), every product split-bf16: per hidden layer dZ.(alpha W) [K x CIN], A.Q
        # [CIN x CIN], T = dZ^T A [K x CIN], the upper triangle of G = A^T A; conv5: A4.Q [128 x 128] + 10 of G's 16 tiles
        hidden = sum(2.0 * r * (k * c + c * c + k * c + c * c * 3 / 4) for k, c in ((128, 64), (64, 64), (64, 64)))
        top = 2.0 * r * (128 * 128 + 128 * 128 * 10 / 16)
        plain = 2.0 * 2.0 * r * (64 * 64 + 64 * 64 + 64 * 128 + 128 * F)  # textbook input + weight gradients
        add_mfma("pointnet_backward[", "pointnet_backward", flops_split=hidden + top, useful=plain,
                 note="Q form: issues A.Q and Gram products instead of re-reading the layers' outputs; `fp32_equiv_tflops` "
                      "counts the textbook 2 x forward FLOP")
        # the bf16 variant is bound by HBM: bytes of the bf16 activations it must move (written once forward; read by
        # the next layer, by both gradient kernels of its own layer and by the gradient kernels of the next)
        fwd = 2.0 * r * (3 * 64 + 64 * 64 + 64 * 64 + 64 * 128 + 128 * F)
        for prefix, name, flops, byts in (
                ("pointnet_forward_bf16", "pointnet_forward_bf16", fwd, 2.0 * r * (2 * (64 * 3) + 2 * 128 + 2 * F) + 12.0 * r),
                ("pointnet_backward_bf16", "pointnet_backward_bf16", 2.0 * fwd, 2.0 * r * (2 * F + 2112))):
            ms = per_step(prefix)
            if ms is not None:
                secs = ms * 1e-3
                rows[name] = {"ms_per_step": ms, "bound": "hbm", "frac": byts / secs / 1e9 / HBM_PEAK_GBS,
                              "peak": "8 TB/s HBM (bf16 activation bytes that must move / time)", "hbm_bytes": byts,
                              "mfma_bf16_frac": flops / secs / MFMA_BF16_PEAK}
    else:
        gemm = 2.0 * nv * N * (3 * 128 + 64 * 128 + 64 * 256 + 128 * 512 + 512 * F)
        add_mfma("dgcnn_forward", "dgcnn_forward", flops_split=gemm,
                 note="row GEMMs only (split-bf16); the kNN stages have their own `roofline` entry")
        add_mfma("dgcnn_backward", "dgcnn_backward", flops_split=2.0 * gemm)
    if "transformer_layers" in cfg.model:
        D, FF, L, H = F, cfg.model.transformer_feat_dim, cfg.model.transformer_layers, cfg.model.transformer_heads
        M = B * P
        tf = L * (2.0 * M * D * 3 * D + 2.0 * M * D * D + 4.0 * M * D * FF + 4.0 * B * H * P * P * (D // H))
        add_mfma("transformer_forward", "transformer_forward", flops_f32=tf,
                 note="launch / dependency latency bound: 17 launches of 4-13 us (a persistent form loses: LABBOOK 6.3)")
        add_mfma("transformer_backward", "transformer_backward", flops_f32=2.0 * tf)
    # per-part Chamfer (csrc/gate_nn.hip): every pair of a part's two clouds is BOUNDED by 1 / 1024 of a bf16 matrix
    # instruction (32 x 32 x 16: 32 FLOP per pair), then ~1.5 % of the pairs are evaluated with the pinned arithmetic
    ms = per_step("assembly_part_chamfer")
    if ms is not None:
        secs = ms * 1e-3
        pairs = 2.0 * nv * N * N
        alg = 24.0 * 2 * nv * N
        rows["assembly_part_chamfer"] = {
            "ms_per_step": ms, "bound": "issue (v_min3 trees of the bound + the answer phase)",
            "frac": 32.0 * pairs / secs / MFMA_BF16_PEAK,
            "peak": "2.5 PFLOP/s bf16 matrix rate (the bound's Gram products: 32 FLOP per pair)",
            "pairs_bounded_per_step": pairs, "hbm_bytes": alg, "hbm_frac": alg / secs / 1e9 / HBM_PEAK_GBS,
            "exhaustive_equivalent": {"lane_ops": 8.6 * pairs, "seconds_at_valu_peak": 8.6 * pairs / VALU_PEAK_LANE_OPS,
                                      "speedup_over_valu_peak_scan": 8.6 * pairs / VALU_PEAK_LANE_OPS / secs,
                                      "note": "what the exhaustive scan it replaced would need at the full VALU issue rate: "
                                              "a speed-up figure, NOT a utilisation"}}
    return rows


# ---- the drop-in operator on its own (BASELINE.json metric: "+ Chamfer kernel GB/s") ----------------------------------------
def chamfer_standalone(dev, reps=20):
    """mpa_chamfer_forward through the C ABI (multi_part_assembly_amd.chamfer.chamfer_forward, the `chamfer_cuda`
    replacement) at SURVEY.md §8(d)'s two standalone shapes, timed per call with HIP events on the launch stream:
    the per-part call [640, 1000, 3]^2 of rot_points_cd_loss and the whole-shape call [32, 20000, 3]^2 of shape_cd_loss
    with the 1e3 padding fill applied (utils/loss.py:173-199), the latter for an untrained (far) and a trained (close)
    prediction.  GB/s on §8(d)'s 24 algorithmic bytes per point of both clouds; the exhaustive scan of the same call
    is timed beside the pruned search."""
    import torch
    from multi_part_assembly_amd import _lib, chamfer, synthetic
    from multi_part_assembly_amd.transforms import pose_apply

    batch = synthetic.make_batch(BATCH, PARTS, POINTS, preset="everyday", seed=1234, device=dev)
    v = batch["part_valids"]
    pts = batch["part_pcs"]
    g = torch.Generator(device="cpu").manual_seed(99)
    q_far = torch.nn.functional.normalize(torch.randn(BATCH, PARTS, 4, generator=g), dim=-1).to(dev)
    t_far = (torch.rand(BATCH, PARTS, 3, generator=g) * 0.8 - 0.4).to(dev)
    q_gt = torch.where(v[..., None] > 0, batch["part_quat"], q_far.new_tensor([1.0, 0.0, 0.0, 0.0]))
    q_near = torch.nn.functional.normalize(q_gt + 0.02 * torch.randn(BATCH, PARTS, 4, generator=g).to(dev), dim=-1)
    t_near = batch["part_trans"] + 0.01 * torch.randn(BATCH, PARTS, 3, generator=g).to(dev)

    def shape(q, t):  # shape_cd_loss's cloud: padded parts := 1e3, then the pose, flattened to [B, P*N, 3]
        return pose_apply(pts, q, t, mask=v, fill=1e3).reshape(BATCH, PARTS * POINTS, 3).contiguous()

    gt = shape(q_gt, batch["part_trans"])
    cases = [
        ("[640,1000,3]^2 per-part clouds (rot_points_cd_loss: rotated by prediction vs ground truth)",
         pose_apply(pts, q_far).reshape(BATCH * PARTS, POINTS, 3).contiguous(),
         pose_apply(pts, q_gt).reshape(BATCH * PARTS, POINTS, 3).contiguous()),
        ("[32,20000,3]^2 whole shapes, 1e3 padding fill, untrained prediction (random poses)", shape(q_far, t_far), gt),
        ("[32,20000,3]^2 whole shapes, 1e3 padding fill, trained prediction (poses within 2 % of ground truth)",
         shape(q_near, t_near), gt),
    ]

    def timed(a, b, variant, n):
        for _ in range(3):
            chamfer.chamfer_forward(a, b, variant=variant)
        timer = _lib.KernelTimer()
        _lib.KernelTimer.active = timer
        try:
            for _ in range(n):
                out = chamfer.chamfer_forward(a, b, variant=variant)
            torch.cuda.synchronize()
        finally:
            _lib.KernelTimer.active = None
        (rec,) = timer.summary().values()
        return rec["avg_ms"], rec["launches"], out

    rows = []
    for name, a, b in cases:
        B_, n1, n2 = a.shape[0], a.shape[1], b.shape[1]
        ms, launches, out = timed(a, b, None, reps)
        alg = 24.0 * B_ * (n1 + n2)
        pairs = 2.0 * B_ * n1 * n2
        row = {"case": name, "calls": launches, "avg_call_ms": ms, "algorithmic_bytes_per_call": alg,
               "GBps": alg / (ms * 1e-3) / 1e9, "hbm_frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
               "exhaustive_pair_evals_per_call": pairs}
        pruned = n1 * n2 >= 9_000_000
        gated = not pruned and min(n1, n2) >= 192  # (chamfer.hip: MPA_CHAMFER_GATE_MIN)
        row["search"] = ("grid-pruned exact (sort + search + hand-back scan, 3 launches)" if pruned else
                         "matrix-core gated exact (gate_nn.hip: one bf16 MFMA per 32 x 32 pairs bounds them, the pinned "
                         "arithmetic answers; 1 launch)" if gated else "exhaustive scan")
        if pruned or gated:
            ms2, _, out2 = timed(a, b, 2, 3)
            row["exhaustive_scan_ms"] = ms2
            row["exhaustive_scan_valu_frac"] = 8.6 * pairs / (ms2 * 1e-3) / VALU_PEAK_LANE_OPS
            row["bit_equal_to_exhaustive_scan"] = all(bool(torch.equal(x, y)) for x, y in zip(out, out2))
        else:
            row["valu_frac"] = 8.6 * pairs / (ms * 1e-3) / VALU_PEAK_LANE_OPS
        rows.append(row)
    return {"entry": "mpa_chamfer_forward (include/mpa_hip.h) via multi_part_assembly_amd.chamfer.chamfer_forward",
            "timing": "HIP events on the launch stream around each call (workspace allocation outside the events)",
            "bytes_rule": "SURVEY.md 8(d): 24 B per point of both clouds (12 B xyz read, 4 B distance + 8 B int64 index written)",
            "cases": rows}


def main():
    global N_BATCHES
    args = parse_args()
    N_BATCHES = max(1, args.batches)
    if args.gpus > 1 and "RANK" not in os.environ:
        relaunch_distributed(args)
    # ONE line on stdout, whatever the libraries print: RCCL writes a five-line version banner to stdout when the first
    # communicator comes up.  File descriptor 1 points at stderr for the whole run; the JSON line goes to the real stdout.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = "RANK" in os.environ and "MASTER_ADDR" in os.environ  # launched by torch.distributed.run
    backend = dp_backend() if distributed else None
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "gloo":  # ranks share the visible GPUs
            local_rank = local_rank % max(1, torch.cuda.device_count())
            torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from multi_part_assembly_amd import _lib
    from multi_part_assembly_amd.pn_transformer import build_model
    from multi_part_assembly_amd.trainer import Trainer

    cfg, batch, desc, B, P = workload(args.config, rank, dev)
    batches = [batch] + [workload(args.config, rank, dev, k)[1] for k in range(1, N_BATCHES)]
    torch.manual_seed(0)  # same initial weights on every rank (and broadcast from rank 0 anyway)
    model = build_model(cfg).to(dev)
    if args.dtype == "bf16":
        from multi_part_assembly_amd.encoder import PointNet
        nets = [m for m in model.modules() if isinstance(m, PointNet)]
        if not nets:
            sys.exit("bench.py: --dtype bf16 needs a PointNet encoder (configs c1, c2, c4)")
        for m in nets:
            m.precision = "bf16"
    use_graph = args.graph and not args.eager
    trainer = Trainer(model, cfg, use_graph=use_graph)
    use_graph = bool(trainer.use_graph)  # (semantic-matching models refuse the capture and run eager: the line says what ran)
    parts_of = [bt.pop("num_parts") for bt in batches]
    num_parts = parts_of[0]
    valid_per_batch = [int(sum(n)) for n in parts_of]
    # units of the roofline kernels PER LAUNCH, averaged over the launches of the timed region (batch i % N_BATCHES in step i)
    used = [valid_per_batch[i % N_BATCHES] for i in range(max(1, args.steps))]
    valid_parts = sum(used) / len(used)

    # A generation-2 collection (tens of ms with torch loaded) inside K ~2 ms steps would be the measurement, so the collector
    # is emptied and switched off — BEFORE the warm-up steps: the same collection between warm-up and the timed region left
    # the GPU idle for those tens of ms, its clocks fell, and the first timed steps ran like the first steps of a process
    # (tools/exp_step_profile.py: 3.07, 2.37, 2.33, 2.30 ms, then 2.26) — ~1 ms on top of a 20-step region.
    gc.collect()
    gc.disable()
    # untimed: W warm-up steps (+ the eager settle steps and the capture itself in graph mode)
    for i in range(args.warmup + (trainer.graph_warmup + 1 if use_graph else 0)):
        trainer.train_step(batches[i % N_BATCHES], i)

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    # Inside the timed region only the roofline kernel is bracketed by HIP events (recorded by the library right around
    # it: two records per step); the per-phase table of every entry point (~60 records per step, 0.1 ms of gaps in a
    # 2.7 ms step) is taken in a second, untimed pass over the same K steps.
    roof_timer = _lib.KernelTimer(only=("shape_search_kernel", "dgcnn_knn") + (("chamfer_forward[",) if args.config == "c1" else ()))
    if not use_graph:
        _lib.KernelTimer.active = roof_timer
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = trainer.train_step(batches[i % N_BATCHES], i)
    fence()
    elapsed = time.perf_counter() - t0
    gc.enable()
    _lib.KernelTimer.active = None
    final_loss = float(loss)
    self_check = None
    if args.self_check > 0:  # does the K-step mean hold over a longer run?  (same fences, same batch)
        gc.collect()
        gc.disable()
        fence()
        t1 = time.perf_counter()
        for i in range(args.self_check):
            trainer.train_step(batches[i % N_BATCHES], i)
        fence()
        long_ms = 1e3 * (time.perf_counter() - t1) / args.self_check
        gc.enable()
        short_ms = 1e3 * elapsed / max(1, args.steps)
        self_check = {"steps": args.self_check, "ms_per_step": long_ms, "ratio_to_timed_mean": long_ms / short_ms,
                      "holds": abs(long_ms / short_ms - 1.0) < 0.05}
    collectives = None
    if distributed:
        from multi_part_assembly_amd.dp import measure_collectives
        collectives = measure_collectives(trainer, batch, steps=min(args.steps, 10), fence=fence)
        if collectives is not None and collectives["local_ms_per_step"] is not None:
            dp_ms = 1e3 * elapsed / max(1, args.steps)
            hidden = sum(collectives["allreduce_ms"]) - max(0.0, dp_ms - collectives["local_ms_per_step"])
            collectives["exposed_ms"] = max(0.0, dp_ms - collectives["local_ms_per_step"])
            collectives["overlap_frac"] = (max(0.0, min(1.0, hidden / sum(collectives["allreduce_ms"])))
                                           if sum(collectives["allreduce_ms"]) > 0 else None)
    timer = _lib.KernelTimer()
    if rank == 0 and not distributed:  # (with several ranks the backward hooks of this pass would start collectives)
        # per-phase timing pass (not timed): the same K steps launched eagerly with every entry point instrumented
        _lib.KernelTimer.active = timer
        for i in range(args.steps):
            trainer._fwd_bwd(batches[i % N_BATCHES])
            trainer.optimizer.prepare_hyper()
            trainer.optimizer.step_dev()
        torch.cuda.synchronize()
        _lib.KernelTimer.active = None
    if distributed:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / max(1, args.steps)
        value = world * B * P * args.steps / elapsed
        kernels = timer.summary()
        if not use_graph:
            kernels.update(roof_timer.summary())  # the roofline kernel's entry: its in-region measurement
        N = POINTS
        timing = ("HIP events recorded by libmpa_hip.so right before/after the kernel on its launch stream, "
                  + ("in an eager pass over the same K steps right after the timed graph replays"
                     if use_graph else "inside the timed region (the other entries of `kernels` come from a second, "
                     "untimed pass over the same K steps with every entry point instrumented)"))
        roofline = None
        if cfg.model.encoder == "dgcnn":
            # kNN graph of the first EdgeConv stage (C = 3): SURVEY.md §8d — n*N*(4C read + 20*8 index write)
            hit = _find(kernels, "dgcnn_knn[")
            if hit:
                per_layer = {}
                for name, k in hit:
                    C = int(name.split("C=")[1].rstrip("]"))
                    secs = k["avg_ms"] * 1e-3
                    alg = float(valid_parts) * N * (4 * C + 20 * 8)
                    pairs = float(valid_parts) * N * N
                    per_layer[f"C={C}"] = {
                        "avg_launch_ms": k["avg_ms"], "launches": k["launches"], "algorithmic_bytes": alg,
                        "hbm_GBps": alg / secs / 1e9, "hbm_frac": alg / secs / 1e9 / HBM_PEAK_GBS,
                        "score_flops": pairs * (2 * C + 1), "score_tflops": pairs * (2 * C + 1) / secs / 1e12,
                        "pairs_per_s": pairs / secs}
                name, k = max(hit, key=lambda kv: kv[1]["total_ms"])
                C = int(name.split("C=")[1].rstrip("]"))
                secs = k["avg_ms"] * 1e-3
                alg = float(valid_parts) * N * (4 * C + 20 * 8)
                pairs = float(valid_parts) * N * N
                traffic, traffic_src = None, None
                pmc = sorted((ROOT / "profiles").glob("r*_pmc_knn_kernel_c5.json" if args.config == "c5"
                                                     else "r*_pmc_knn_kernel.json"))
                if pmc:  # HBM-side bytes per launch from the committed rocprofv3 --pmc passes OF THIS WORKLOAD (same
                    rec = json.loads(pmc[-1].read_text())  # clouds per launch), else null
                    same = rec.get("clouds_per_launch") is not None and abs(rec["clouds_per_launch"] - valid_parts) < 0.5
                    if same and str(C) in rec.get("per_width", {}):
                        traffic = rec["per_width"][str(C)]["traffic_bytes_per_launch"]
                        traffic_src = f"profiles/{pmc[-1].name}: {rec['correction']}"
                roofline = {
                    "kernel": (f"dg::knn_wide (rownorm_kernel + knn_split_kernel + knn_gram_kernel bound / collect + knn_rerank_kernel: k = 20 "
                               f"nearest neighbours in {C}-d feature space, {valid_parts} clouds of {N} points)"
                               if C >= 64 else f"dg::knn3_kernel (k = 20 nearest neighbours of {valid_parts} clouds of {N} "
                               f"points in 3-d)"),
                    "bound": "hbm", "achieved": alg / secs / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": alg / secs / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                    "avg_launch_ms": k["avg_ms"], "launches": k["launches"], "algorithmic_bytes_per_launch": alg,
                    "binding": {"bound": "mfma_bf16" if C >= 64 else "valu",
                                "note": "exact top-20 as a shortlist search (csrc/dg_knn_fast.h): two passes (bound + "
                                        "collect) of ONE-product bf16 Gram tiles over the centred rows (4 C FLOP per pair) "
                                        "on the bf16 matrix cores, then the pinned fp32 chain for the survivors (~25 per "
                                        "query); the survivors' row gathers through the L2 and the tiles' staging bind "
                                        "it, neither HBM nor the MFMA rate does (SURVEY.md §7 hard part 1)" if C >= 64 else
                                        "exhaustive exact top-20 on the VALU (12 lane-slots per pair)",
                                "gram_flops_per_launch": pairs * 4.0 * C if C >= 64 else None,
                                "frac": (pairs * 4.0 * C / secs / MFMA_BF16_PEAK) if C >= 64 else
                                        (pairs * 12.0 / secs / VALU_PEAK_LANE_OPS),
                                "frac_of_fp32_mfma_if_exhaustive": (pairs * (2 * C + 1) / secs / MFMA_F32_PEAK)
                                if C >= 64 else None},
                    "per_layer": per_layer, "timing": timing}
        else:
            dom = _find(kernels, "shape_search_kernel[")
            phase = _find(kernels, "assembly_shape_chamfer[")
            if dom:
                k = dom[0][1]
                # 24 B per point of both clouds (12 B xyz read, 4 B distance, 8 B index written);
                # padded parts cost nothing here (one representative point each), so valid points only.
                alg_bytes = 2.0 * 24.0 * valid_parts * N
                brute_pairs = 2.0 * N * N * sum(sum(n * n for n in parts_of[i % N_BATCHES])
                                                for i in range(max(1, args.steps))) / max(1, args.steps)
                secs = k["avg_ms"] * 1e-3
                achieved = alg_bytes / secs / 1e9
                traffic, traffic_src = None, None
                pmc = sorted((ROOT / "profiles").glob("r*_pmc_dominant_kernel.json"))
                if pmc:  # HBM-side bytes per launch of this kernel from the committed rocprofv3 --pmc passes of c2
                    rec = json.loads(pmc[-1].read_text())
                    if (args.config in ("c2", "c4") and rec.get("clouds_per_launch") is not None
                            and abs(rec["clouds_per_launch"] - valid_parts) < 0.5):
                        traffic, traffic_src = rec["traffic_bytes_per_launch"], f"profiles/{pmc[-1].name}: {rec['correction']}"
                from multi_part_assembly_amd.loss import search_mode, SEARCH_MODES
                mode = {v: k for k, v in SEARCH_MODES.items()}[search_mode(cfg.loss.get("shape_search", None))]
                kname = {"grid": "mpa::grid_search_kernel (exact grid-pruned whole-shape Chamfer search of the fused loss, both "
                                 "directions; cfg.loss.shape_search / MPA_SHAPE_SEARCH = leaf | auto: mpa::leaf_search_kernel)",
                         "leaf": "mpa::leaf_search_kernel<true> + leaf_search_heavy_kernel<true> (exact whole-shape Chamfer "
                                 "search of the fused loss over the parts' k-d leaves, both directions)",
                         "auto": "mpa::leaf_search_kernel<true> + leaf_search_heavy_kernel<true> + grid_search_kernel (exact "
                                 "whole-shape Chamfer search of the fused loss, routed per sample)",
                         "brute": "assembly_nn_kernel<SHAPE> (exhaustive whole-shape Chamfer scan of the fused loss)"}[mode]
                bnote = {"grid": "exact pruned search: dependent latency and VALU issue of short candidate lists bind it "
                                 "(SQ counters: ~40 % of wave cycles waiting, ~40 % issuing), neither HBM nor the MFMA rate "
                                 "does (DESIGN.md §4); an exhaustive scan of the same points would be VALU-bound",
                         "brute": "exhaustive scan: VALU-bound"}.get(mode,
                         "exact pruned search: VALU issue slots of the leaf selection, the per-lane box tests and the "
                         "reduction of the matrix cores' 32 x 64 gate tiles bind it, neither HBM nor the MFMA rate does "
                         "(DESIGN.md §4); an exhaustive scan of the same points would be VALU-bound")
                roofline = {
                    "kernel": kname, "shape_search": mode,
                    "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                    "avg_launch_ms": k["avg_ms"], "launches": k["launches"],
                    "algorithmic_bytes_per_launch": alg_bytes,
                    "binding": {"bound": "latency / valu-issue" if mode == "grid" else "valu-issue",
                                "note": bnote,
                                "equivalent_brute_force_pair_evals_per_s": brute_pairs / secs,
                                "valu_frac_if_brute_force": 8.6 * brute_pairs / secs / VALU_PEAK_LANE_OPS},
                    "oracle_note": "quaternion algebra of the loss restated from pytorch3d (un-vendored): parity "
                                   "unpinned at that boundary",
                    "timing": timing,
                    "whole_phase_avg_ms": phase[0][1]["avg_ms"] if phase else None,
                }
            elif args.config == "c1":
                # (only with MPA_FUSED_SEMANTIC=0: since round 6 the semantic models run the fused loss and take the branch above)
                # plumbing case (P = 2: no grid phase): the loss runs the drop-in operator itself, 5 min-of-N samples x
                # (whole-shape call [B, P*N, 3]^2 + per-part call) per step; the whole-shape call dominates
                hit = _find(kernels, f"chamfer_forward[{B}x{P * N}x{P * N}]")
                if hit:
                    k = hit[0][1]
                    secs = k["avg_ms"] * 1e-3
                    alg_bytes = 24.0 * B * 2 * P * N
                    pairs = 2.0 * B * (P * N) ** 2
                    roofline = {
                        "kernel": f"chamfer_nn_kernel behind mpa_chamfer_forward, whole-shape call [{B}, {P * N}, 3]^2 of "
                                  "shape_cd_loss (exhaustive scan: below the grid-pruned search's size threshold)",
                        "bound": "hbm", "achieved": alg_bytes / secs / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": alg_bytes / secs / 1e9 / HBM_PEAK_GBS, "traffic": None, "traffic_source": None,
                        "avg_launch_ms": k["avg_ms"], "launches": k["launches"],
                        "algorithmic_bytes_per_launch": alg_bytes,
                        "binding": {"bound": "valu", "pair_evals_per_launch": pairs,
                                    "frac": 8.6 * pairs / secs / VALU_PEAK_LANE_OPS,
                                    "note": "4 samples x 2 directions x 2000 queries = 63 waves of work on 1024 SIMDs: the "
                                            "call is far too small to fill the chip (launch-latency territory)"},
                        "timing": timing}
        rccl = None
        if distributed and backend == "nccl":
            try:
                rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:  # noqa: BLE001
                rccl = "unknown"
        line = {
            "metric": "train-step parts/sec (BxP) at N=1000 pts",
            "value": value, "unit": "parts/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": desc, "name": args.config,
                       "per_gpu_batch": B, "max_parts": P, "points_per_part": POINTS,
                       "valid_parts_rank0": valid_parts, "valid_parts_per_batch_rank0": valid_per_batch,
                       "batches": f"{N_BATCHES} distinct synthetic batches (seeds 1234 + rank + 1000 k), batch i % {N_BATCHES} "
                                  "in step i of warm-up and timed region alike; part counts redrawn until their sum is within "
                                  "4 of its expectation (352 everyday / 512 artifact) except the historical batch 0 of rank 0",
                       "parallelism": f"dp{world}",
                       "rccl_ranks": world if distributed and backend == "nccl" else 0, "rccl_version": rccl,
                       "dp_backend": backend,
                       "launch": "hip-graph replay" if use_graph else "eager",
                       **({"precision_note": "performance variant: PointNet encoder with bf16 stored activations and "
                           "v_mfma_f32_32x32x16_bf16 GEMMs, fp32 BatchNorm statistics and gradients; everything else "
                           "fp32; parity is claimed for the f32 line only"} if args.dtype == "bf16" else {})},
            "final_loss": final_loss,
            "kernels": kernels,
            "roofline": roofline,
            "kernel_table": kernel_table(kernels, valid_parts, cfg, B, P),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.config, args.cpu_batch, dev)
        if world == 1 and not args.no_chamfer_standalone:
            line["chamfer_standalone"] = chamfer_standalone(dev)
        if self_check is not None:
            line["self_check"] = self_check
        if collectives is not None:
            line["collectives"] = collectives
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
