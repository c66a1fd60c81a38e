"""Data-parallel training over the GPUs of one node: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests), gradient MEAN across ranks each step.

The reference reaches this through Lightning's `strategy='ddp'` (scripts/train.py:81-95,141): DDP
wraps the module, buckets gradients at 25 MB and all-reduces them during backward; BatchNorm
statistics stay per rank (no SyncBN) and the learning rate is not scaled.  Same semantics here,
different mechanism: all gradients already live in ONE flat buffer (optim.FlatBuffers), split into
a few contiguous buckets ordered by when backward finishes them (pose head + transformer first, the
point encoder last).  Each bucket is all-reduced with a single collective as soon as its last
gradient has been accumulated, overlapping the remaining backward; the 1/world_size of the mean is
folded into the fused Adam kernel.  At 13 MB of gradients the collective is latency-bound
(SURVEY.md §5.8), so few large messages beat DDP's generic bucket walk.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .optim import FlatBuffers


class BucketedGradReducer:
    """Async all-reduce (SUM) of contiguous slices of a flat gradient buffer.

    `bucket_sizes`: number of parameters per bucket, in FlatBuffers order; a bucket's collective is
    issued from the post-accumulate hook of whichever of its parameters receives its gradient last.
    """

    def __init__(self, flat: FlatBuffers, bucket_sizes, group=None):
        assert sum(bucket_sizes) == len(flat.params)
        self.flat, self.group = flat, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.buckets, first = [], 0
        for size in bucket_sizes:
            if size == 0:
                continue
            start, end = flat.span(first, first + size - 1)
            self.buckets.append({"slice": flat.flat_grad[start:end], "count": size, "ready": 0})
            first += size
        self.pending = []
        self.enabled = True  # False: hooks and finish() issue no collective (bench.py: the step time without them)
        self._seen = set()
        self._owner = {}
        idx = 0
        for b, bucket in enumerate(self.buckets):
            for _ in range(bucket["count"]):
                p = flat.params[idx]
                self._owner[p] = b
                if self.world > 1:
                    p.register_post_accumulate_grad_hook(self._on_grad)
                idx += 1

    def _on_grad(self, param):
        """Gradient of `param` is final for this step.  Idempotent per step: a parameter whose gradient a HIP
        backward kernel wrote directly is announced by the GradSink, and — depending on the torch version — also by
        autograd's post-accumulate hook, which fires even when the Function returned None for that input."""
        if id(param) in self._seen or not self.enabled:
            return
        self._seen.add(id(param))
        bucket = self.buckets[self._owner[param]]
        bucket["ready"] += 1
        if bucket["ready"] == bucket["count"]:
            self.pending.append(dist.all_reduce(bucket["slice"], op=dist.ReduceOp.SUM,
                                                group=self.group, async_op=True))

    def finish(self):
        """Issue collectives for buckets whose hooks did not all fire (parameters unused in this
        step), wait for everything, reset.  Returns the scale (1/world) still to be applied."""
        if self.world > 1 and self.enabled:
            for bucket in self.buckets:
                if bucket["ready"] != bucket["count"]:
                    self.pending.append(dist.all_reduce(bucket["slice"], op=dist.ReduceOp.SUM,
                                                        group=self.group, async_op=True))
                bucket["ready"] = 0
            for work in self.pending:
                work.wait()
            self.pending.clear()
        for bucket in self.buckets:
            bucket["ready"] = 0
        self._seen.clear()
        return 1.0 / self.world


def broadcast_from_rank0(flat: FlatBuffers, module: torch.nn.Module, group=None):
    """Start every rank from rank 0's parameters and buffers (what DDP does at construction)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    dist.broadcast(flat.flat_param, src=0, group=group)
    for buf in module.buffers():
        dist.broadcast(buf, src=0, group=group)


def bucket_sizes_for(model, flat: FlatBuffers):
    """Two buckets in backward-completion order: [everything but the encoder, the encoder]."""
    enc = {id(p) for p in getattr(model, "encoder").parameters()} if hasattr(model, "encoder") else set()
    n_enc = sum(1 for p in flat.params if id(p) in enc)
    return [len(flat.params) - n_enc, n_enc]


def ordered_parameters(model):
    """Parameters grouped so that each gradient bucket is contiguous: non-encoder first."""
    enc = {id(p) for p in model.encoder.parameters()} if hasattr(model, "encoder") else set()
    rest = [p for p in model.parameters() if id(p) not in enc and p.requires_grad]
    return rest + [p for p in model.parameters() if id(p) in enc and p.requires_grad]


def measure_collectives(trainer, batch, steps=10, fence=None, reps=10):
    """Where a data-parallel step's time goes (bench.py, after its timed region; every rank must call this): per
    gradient bucket its size and the time of its all-reduce ALONE (nothing else running: `reps` back-to-back calls
    between fences), and the step time with the collectives switched off.  Together with the measured data-parallel
    step time: exposed = dp - local, overlap = 1 - exposed / sum(all-reduce).  The replicas diverge during the
    collective-free steps — call this last."""
    import time

    reducer = trainer.reducer
    if reducer is None or not dist.is_initialized():
        return None
    on_gpu = trainer.flat.flat_grad.is_cuda

    def sync():
        if fence is not None:
            fence()
        else:
            if on_gpu:
                torch.cuda.synchronize()
            dist.barrier(group=reducer.group)
            if on_gpu:
                torch.cuda.synchronize()

    bucket_bytes, allreduce_ms = [], []
    for bucket in reducer.buckets:
        bucket_bytes.append(int(bucket["slice"].numel() * bucket["slice"].element_size()))
        scratch = torch.zeros_like(bucket["slice"])
        dist.all_reduce(scratch, group=reducer.group)  # warm-up (connection setup on the first call)
        sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            dist.all_reduce(scratch, group=reducer.group)
        sync()
        allreduce_ms.append(1e3 * (time.perf_counter() - t0) / reps)
    local_ms = float("nan")
    if reducer.enabled:  # (graph mode issues its collectives itself: no collective-free variant of its step)
        reducer.enabled = False
        try:
            trainer.train_step(batch, 0)
            sync()
            t0 = time.perf_counter()
            for i in range(steps):
                trainer.train_step(batch, i)
            sync()
            local_ms = 1e3 * (time.perf_counter() - t0) / steps
        finally:
            reducer.enabled = True
    # max over ranks, like the step time itself
    vals = torch.tensor(allreduce_ms + [local_ms], dtype=torch.float64, device=trainer.flat.flat_grad.device)
    dist.all_reduce(vals, op=dist.ReduceOp.MAX, group=reducer.group)
    vals = vals.tolist()
    return {"world": reducer.world, "bucket_bytes": bucket_bytes, "allreduce_ms": vals[:-1],
            "local_ms_per_step": vals[-1] if vals[-1] == vals[-1] else None, "algorithm_note": "ring all-reduce moves 2 (n-1)/n of the bucket per link"}
