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
        if id(param) in self._seen:
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
        if self.world > 1:
            for bucket in self.buckets:
                if bucket["ready"] != bucket["count"]:
                    self.pending.append(dist.all_reduce(bucket["slice"], op=dist.ReduceOp.SUM,
                                                        group=self.group, async_op=True))
                bucket["ready"] = 0
            for work in self.pending:
                work.wait()
            self.pending.clear()
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
