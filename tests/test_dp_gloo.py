"""Data-parallel path on CPU: 2 processes over gloo.  The bucketed flat-gradient all-reduce must
reproduce a single-process emulation that computes each shard's gradients separately and averages
them (the DDP semantics of the reference, scripts/train.py:85,141 — SURVEY.md §8e)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from multi_part_assembly_amd.dp import (BucketedGradReducer, broadcast_from_rank0, bucket_sizes_for,
                                        ordered_parameters)
from multi_part_assembly_amd.optim import FlatBuffers, cosine_warmup_lr


class Toy(nn.Module):
    """Encoder with BatchNorm (per-rank statistics, like the reference: no SyncBN) + head."""

    def __init__(self):
        super().__init__()
        self.encoder = nn.Sequential(nn.Linear(3, 16), nn.BatchNorm1d(16), nn.ReLU(), nn.Linear(16, 8))
        self.head = nn.Linear(8, 2)
        self.unused = nn.Parameter(torch.zeros(5))  # never receives a gradient

    def forward(self, x):
        return self.head(self.encoder(x)).square().mean()


def _shard(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.randn(12, 3, generator=g)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(rank)  # different init per rank: the broadcast must fix that
    model = Toy()
    flat = FlatBuffers(ordered_parameters(model))
    broadcast_from_rank0(flat, model)
    reducer = BucketedGradReducer(flat, bucket_sizes_for(model, flat))
    results = []
    for step in range(2):
        flat.zero_grad()
        model(_shard(rank) + step).backward()
        scale = reducer.finish()
        results.append((flat.flat_grad * scale).clone())
        with torch.no_grad():
            flat.flat_param -= 0.1 * flat.flat_grad * scale
    # the collective break-down bench.py reports with several ranks (dp.measure_collectives), on a stand-in trainer
    from multi_part_assembly_amd.dp import measure_collectives

    class ToyTrainer:
        pass

    tr = ToyTrainer()
    tr.reducer, tr.flat = reducer, flat

    def train_step(batch, i):
        flat.zero_grad()
        model(batch).backward()
        reducer.finish()

    tr.train_step = train_step
    coll = measure_collectives(tr, _shard(rank), steps=2, reps=2)
    assert reducer.enabled
    torch.save({"grads": results, "param": flat.flat_param.clone(), "collectives": coll,
                "offsets": flat.offsets}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_gloo_matches_single_process_emulation(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(tmp_path / f"rank{r}.pt") for r in range(world)]
    # every rank ends with identical parameters and saw identical averaged gradients
    assert torch.equal(outs[0]["param"], outs[1]["param"])
    for a, b in zip(outs[0]["grads"], outs[1]["grads"]):
        assert torch.equal(a, b)

    # the multi-rank bench fields: two buckets (head: 2 + 5 unused + 16 = 23 floats ... in FlatBuffers' padded layout),
    # a stand-alone all-reduce time per bucket and the collective-free step time, identical on both ranks (MAX-reduced)
    for o in outs:
        c = o["collectives"]
        assert c["world"] == 2 and len(c["bucket_bytes"]) == 2 == len(c["allreduce_ms"])
        assert sum(c["bucket_bytes"]) == outs[0]["grads"][0].numel() * 4
        assert all(t > 0 for t in c["allreduce_ms"]) and c["local_ms_per_step"] > 0
    assert outs[0]["collectives"] == outs[1]["collectives"]

    # single-process emulation: rank-0 init, per-shard BN statistics, mean of shard gradients
    torch.manual_seed(0)
    replicas = [Toy() for _ in range(world)]
    replicas[1].load_state_dict(replicas[0].state_dict())
    flats = [FlatBuffers(ordered_parameters(m)) for m in replicas]
    for step in range(2):
        for r in range(world):
            flats[r].zero_grad()
            replicas[r](_shard(r) + step).backward()
        mean = (flats[0].flat_grad + flats[1].flat_grad) / world
        np.testing.assert_allclose(outs[0]["grads"][step].numpy(), mean.numpy(), rtol=1e-6, atol=1e-7)
        with torch.no_grad():
            for f in flats:
                f.flat_param -= 0.1 * mean
    np.testing.assert_allclose(outs[0]["param"].numpy(), flats[0].flat_param.numpy(), rtol=1e-6, atol=1e-7)


def test_flat_buffers_alias_parameters_and_gradients():
    model = Toy()
    flat = FlatBuffers(ordered_parameters(model))
    # non-encoder parameters come first, so the two gradient buckets are contiguous
    sizes = bucket_sizes_for(model, flat)
    assert sizes == [3, 6] and sum(sizes) == len(flat.params)
    model(torch.randn(4, 3)).backward()
    for p, off in zip(flat.params, flat.offsets):
        assert p.data_ptr() == flat.flat_param[off:].data_ptr()
        assert p.grad.data_ptr() == flat.flat_grad[off:].data_ptr()
        assert off % 4 == 0
    assert flat.flat_grad.abs().sum() > 0
    flat.zero_grad()
    assert all(p.grad.abs().sum() == 0 for p in flat.params)


def test_cosine_schedule_restates_reference_quirks():
    lr = cosine_warmup_lr(400, 20, 1e-3, 1e-5)
    assert lr(0) == 1e-5                       # epoch 0 runs at min_lr (utils/lr.py:68-75)
    assert abs(lr(10) - (1e-5 + (1e-3 - 1e-5) * 10 / 20)) < 1e-12
    assert abs(lr(20) - 1e-3) < 1e-12          # end of warm-up = peak
    assert abs(lr(210) - (1e-5 + (1e-3 - 1e-5) * 0.5)) < 1e-9
    assert lr(399) > 1e-5 and abs(lr(400) - lr(0 + 400 % 400 or 400)) >= 0
    no_warm = cosine_warmup_lr(200, 0, 1e-3, 1e-5)
    assert abs(no_warm(1) - (1e-5 + (1e-3 - 1e-5) * (1 + np.cos(np.pi / 200)) / 2)) < 1e-12


def test_cosine_schedule_matches_reference_scheduler(golden):
    """Every epoch's learning rate of the reference's CosineAnnealingWarmupRestarts (utils/lr.py:26-125, driven as
    base_model.py:407-425 does) for the shipped schedules and a short one, restart included."""
    z = golden("lr_schedule")
    for tag in ("pn400", "dgl200", "short"):
        total, ratio, lr, decay = z[tag + ".cfg"]
        fn = cosine_warmup_lr(int(total), int(total * ratio), float(lr), float(lr / decay))
        mine = np.array([fn(e) for e in range(len(z[tag + ".lr"]))])
        np.testing.assert_allclose(mine, z[tag + ".lr"], rtol=1e-12, atol=0, err_msg=tag)


def test_dropout_salts_are_positions_inside_the_model():
    """The dropout seed of a TransformerEncoder mixes in the encoder's index inside ITS model (module order), not a
    process-global construction counter: a model built second in a process draws the same masks as one built first."""
    from multi_part_assembly_amd import config
    from multi_part_assembly_amd.pn_transformer import build_model
    from multi_part_assembly_amd.transformer import TransformerEncoder

    def salts(model):
        return [m._instance for m in model.modules() if isinstance(m, TransformerEncoder)]

    first = build_model(config.pn_transformer_refine_everyday())
    second = build_model(config.pn_transformer_refine_everyday())
    assert salts(first) == salts(second) == list(range(1, len(salts(first)) + 1)) and len(salts(first)) >= 1
