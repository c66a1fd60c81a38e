"""Minimal training harness standing in for `pl.Trainer` on the hot path (the reference drives
`BaseModel.training_step` through Lightning's automatic-optimisation loop, scripts/train.py:82-120):
zero_grad -> training_step -> backward (gradient buckets all-reduced while it runs) -> fused Adam,
with the per-epoch cosine schedule.  No host synchronisation inside a step: the loss is returned as
a device tensor.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .dp import BucketedGradReducer, broadcast_from_rank0, bucket_sizes_for, ordered_parameters
from .optim import FlatBuffers, FusedAdam, cosine_warmup_lr


class Trainer:
    def __init__(self, model, cfg=None, process_group=None):
        self.model = model
        cfg = cfg if cfg is not None else model.cfg
        opt = cfg.optimizer
        self.flat = FlatBuffers(ordered_parameters(model))
        self.optimizer = FusedAdam(self.flat, lr=opt.lr, weight_decay=opt.weight_decay)
        self.schedule = None
        if opt.lr_scheduler:
            total = cfg.exp.num_epochs
            self.schedule = cosine_warmup_lr(total, int(total * opt.warmup_ratio), opt.lr,
                                             opt.lr / opt.lr_decay_factor)
        self.epoch = 0
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        broadcast_from_rank0(self.flat, model, process_group)
        self.reducer = BucketedGradReducer(self.flat, bucket_sizes_for(model, self.flat), process_group)
        self.set_epoch(0)

    def set_epoch(self, epoch):
        self.epoch = epoch
        if self.schedule is not None:
            self.optimizer.lr = self.schedule(epoch)

    def train_step(self, data_dict, batch_idx=0):
        """One optimiser step on this rank's shard; returns the (detached) scalar loss tensor."""
        self.model.train()
        self.optimizer.zero_grad()
        loss = self.model.training_step(data_dict, batch_idx)
        loss.backward()
        self.optimizer.grad_scale = self.reducer.finish()
        self.optimizer.step()
        return loss.detach()
