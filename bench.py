#!/usr/bin/env python3
"""Benchmark of the hot path: one full training step (forward + losses + backward + gradient
all-reduce + fused Adam) of PNTransformer + PointNet on synthetic Breaking-Bad-"everyday"-like part
clouds, B = 32 per GPU, P = 20, N = 1000 (BASELINE.json configs[1]; weak scaling over GPUs).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` = parts (B x P slots, padded slots included, as the metric is
defined) processed per second by the whole job, inputs resident in HBM before the timed region.
Kernels are launched eagerly (the step is GPU-bound: ~2.3 ms of host time for 3.3 ms of GPU work; `--graph`
replays the captured step as one HIP graph instead and measures ~2.5 % less).  The garbage collector is parked
during the timed steps: one generation-2 collection (~30 ms with torch loaded) inside K = 20 steps of 3 ms was
measured as +1.6 ms per step.  `roofline` describes the dominant kernel (the whole-shape Chamfer search of the
fused loss), timed per launch with HIP events that the library records around it on its launch stream inside the
timed region (with --graph: in an eager pass over the same K steps right after it, events cannot be recorded
inside a replay).  `cpu_baseline` is the oracle's reference-equivalent PyTorch-CPU step
timed on this host (rank 0, N = 1 only) on a bounded sample.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import torch
import torch.distributed as dist

BATCH, PARTS, POINTS = 32, 20, 1000
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_LANE_OPS = 78.6e12  # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz (non-packed fp32 VALU issue)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as one HIP graph (default: eager launches; see trainer.py)")
    ap.add_argument("--eager", action="store_true", help="(the default; accepted for symmetry)")
    ap.add_argument("--cpu-batch", type=int, default=4, help="samples in the CPU-baseline step")
    return ap.parse_args()


def cpu_baseline(cpu_batch):
    """Reference-equivalent CPU training step (oracle/nets.py + oracle/chamfer_ref.c), bounded sample:
    the same workload at B = cpu_batch instead of 32 (cost is linear in B), 1 warm-up + 2 timed steps."""
    from multi_part_assembly_amd import config
    from multi_part_assembly_amd.pn_transformer import build_model
    from oracle import nets as on

    cores = len(os.sched_getaffinity(0))
    threads = max(1, min(64, cores))
    torch.set_num_threads(threads)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    cfg = config.pn_transformer_everyday()
    torch.manual_seed(0)
    model = build_model(cfg)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    params = {k: sd[k].requires_grad_() for k, _ in model.named_parameters()}
    g = torch.Generator().manual_seed(1234)
    B, P, N = cpu_batch, PARTS, POINTS
    num_parts = torch.randint(2, P + 1, (B,), generator=g).tolist()
    valids = torch.zeros(B, P)
    for b, k in enumerate(num_parts):
        valids[b, :k] = 1
    pcs = (torch.rand(B, P, N, 3, generator=g) - 0.5) * 0.3 * valids[..., None, None]
    quat = torch.nn.functional.normalize(torch.randn(B, P, 4, generator=g), dim=-1) * valids[..., None]
    batch = {"part_pcs": pcs, "part_quat": quat, "part_valids": valids,
             "part_trans": (torch.rand(B, P, 3, generator=g) * 0.8 - 0.4) * valids[..., None]}
    state = {}

    def step():
        for p in params.values():
            p.grad = None
        stats = {}
        losses, _ = on.pn_transformer_loss(sd, batch, cfg.model.transformer_layers,
                                           cfg.model.transformer_heads, training=True, stats_out=stats)
        losses["loss"].backward()
        on.adam_step(params, {k: p.grad for k, p in params.items()}, state, lr=cfg.optimizer.lr)
        for k, v in stats.items():
            sd[k] = v

    step()
    t0 = time.perf_counter()
    timed = 2
    for _ in range(timed):
        step()
    dt = (time.perf_counter() - t0) / timed
    return {"value": B * P / dt, "unit": "parts/s", "cores": threads, "kind": "port",
            "sample": f"full train step (fwd+loss+bwd+Adam) of the same model at B={B} (vs 32), P={P}, "
                      f"N={N}; {timed} timed steps after 1 warm-up; {dt:.2f} s/step; torch CPU ops + "
                      f"OpenMP C Chamfer on {threads} threads of {cores} visible"}


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = "RANK" in os.environ and "MASTER_ADDR" in os.environ  # launched by torch.distributed.run
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from multi_part_assembly_amd import _lib, config, synthetic
    from multi_part_assembly_amd.pn_transformer import build_model
    from multi_part_assembly_amd.trainer import Trainer

    cfg = config.pn_transformer_everyday()
    torch.manual_seed(0)  # same initial weights on every rank (and broadcast from rank 0 anyway)
    model = build_model(cfg).to(dev)
    use_graph = args.graph and not args.eager
    trainer = Trainer(model, cfg, use_graph=use_graph)
    batch = synthetic.make_batch(BATCH, PARTS, POINTS, preset="everyday", seed=1234 + rank, device=dev)
    num_parts = batch.pop("num_parts")
    valid_parts = int(sum(num_parts))

    # untimed: W warm-up steps (+ the eager settle steps and the capture itself in graph mode)
    for i in range(args.warmup + (trainer.graph_warmup + 1 if use_graph else 0)):
        trainer.train_step(batch, i)

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    gc.collect()
    gc.disable()  # a generation-2 collection (tens of ms with torch loaded) inside K ~3 ms steps would be the measurement
    fence()
    timer = _lib.KernelTimer()
    if not use_graph:
        _lib.KernelTimer.active = timer
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = trainer.train_step(batch, i)
    fence()
    elapsed = time.perf_counter() - t0
    gc.enable()
    _lib.KernelTimer.active = None
    final_loss = float(loss)
    if use_graph and rank == 0:
        # per-kernel timing pass: the same K steps launched eagerly, library-recorded HIP events
        _lib.KernelTimer.active = timer
        for i in range(args.steps):
            trainer._fwd_bwd(batch)
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
        value = world * BATCH * PARTS * args.steps / elapsed
        kernels = timer.summary()
        dom = f"grid_search_kernel[{BATCH}x{PARTS}x{POINTS}]"
        phase = f"assembly_shape_chamfer[{BATCH}x{PARTS}x{POINTS}]"
        roofline = None
        if dom in kernels:
            k = kernels[dom]
            # algorithmic traffic of the whole-shape search kernel: every valid point of both
            # shapes is read once as a 16 B query record and once as a 16 B target record and produces a 4 B
            # distance and a 4 B index -> 40 B per valid point and shape; padded slots cost nothing.
            alg_bytes = 2.0 * 40.0 * valid_parts * POINTS
            brute_pairs = 2.0 * POINTS * POINTS * sum(n * n for n in num_parts)
            secs = k["avg_ms"] * 1e-3
            achieved = alg_bytes / secs / 1e9
            traffic, traffic_src = None, None
            pmc = sorted((ROOT / "profiles").glob("r*_pmc_dominant_kernel.json"))
            if pmc:  # HBM-side bytes per launch of this kernel from the committed rocprofv3 --pmc passes
                rec = json.loads(pmc[-1].read_text())
                traffic, traffic_src = rec["traffic_bytes_per_launch"], f"profiles/{pmc[-1].name}: {rec['correction']}"
            roofline = {
                "kernel": "mpa::grid_search_kernel (exact grid-pruned whole-shape Chamfer search of the fused "
                          "loss, both directions)",
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_ms": k["avg_ms"], "launches": k["launches"],
                "algorithmic_bytes_per_launch": alg_bytes,
                "timing": "HIP events recorded by libmpa_hip.so right before/after the kernel on its launch stream, "
                          + ("in an eager pass over the same K steps right after the timed graph replays"
                             if use_graph else "inside the timed region"),
                "whole_phase_avg_ms": kernels[phase]["avg_ms"] if phase in kernels else None,
                # the search is VALU-bound, not HBM-bound: pair evaluations an exhaustive scan of
                # the same valid points would need, per second of this kernel
                "equivalent_brute_force_pair_evals_per_s": brute_pairs / secs,
            }
        line = {
            "metric": "train-step parts/sec (BxP) at N=1000 pts",
            "value": value, "unit": "parts/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "pn_transformer + PointNet encoder, Breaking-Bad-everyday-like "
                                   "synthetic clouds, B=32 per GPU, P=20, N=1000, geometric loss, Adam "
                                   "(BASELINE.json configs[1])",
                       "per_gpu_batch": BATCH, "max_parts": PARTS, "points_per_part": POINTS,
                       "valid_parts_rank0": valid_parts, "parallelism": f"dp{world}",
                       "launch": "hip-graph replay" if use_graph else "eager"},
            "final_loss": final_loss,
            "kernels": kernels,
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_batch)
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
