"""Optimiser and LR schedule of the training step.

`FusedAdam` keeps every parameter, gradient and Adam moment of the model in four flat fp32 buffers
and performs the step with one HIP kernel (csrc/adam.hip); the flat gradient buffer is also what the
data-parallel all-reduce operates on (multi_part_assembly_amd/dp.py) — one collective, no per-tensor
bookkeeping.  Replaces torch.optim.Adam/AdamW as configured by the reference
(models/modules/base_model.py:389-406).

`cosine_warmup_lr` restates the per-epoch schedule of the reference's CosineAnnealingWarmupRestarts
(utils/lr.py:26-125) for cycle_mult = gamma = 1, the only way the reference uses it
(base_model.py:411-417).
"""
from __future__ import annotations

import math

import torch

from . import _lib


class FlatBuffers:
    """Re-homes the parameters of a model (and their .grad) as views of two flat fp32 buffers.

    Device-agnostic (the gloo/CPU tests of the data-parallel path use it too).  `order` lets the
    caller group parameters into contiguous regions — the gradient buckets of dp.py."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatBuffers: no trainable parameters")
        dev = self.params[0].device
        offsets, total = [], 0
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise RuntimeError("FlatBuffers: all parameters must be fp32 on one device")
            offsets.append(total)
            total += (p.numel() + 3) // 4 * 4  # keep every tensor 16-byte aligned
        self.numel, self.offsets = total, offsets
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros_like(self.flat_param)
        with torch.no_grad():
            for p, off in zip(self.params, offsets):
                view = self.flat_param[off:off + p.numel()].view_as(p)
                view.copy_(p)
                p.data = view
                p.grad = self.flat_grad[off:off + p.numel()].view_as(p)

    def span(self, first, last):
        """[start, end) element range covering parameters first..last (inclusive indices)."""
        return self.offsets[first], self.offsets[last] + (self.params[last].numel() + 3) // 4 * 4

    def zero_grad(self):
        """One memset; gradients stay views of the flat buffer (never set to None)."""
        self.flat_grad.zero_()


class FusedAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 decoupled_weight_decay=None):
        self.flat = params if isinstance(params, FlatBuffers) else FlatBuffers(params)
        self.params = self.flat.params
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        # the reference switches to AdamW as soon as weight_decay > 0 (base_model.py:394-404)
        self.decoupled = (weight_decay > 0) if decoupled_weight_decay is None else decoupled_weight_decay
        self.step_count = 0
        self.grad_scale = 1.0
        self.numel = self.flat.numel
        self.flat_param, self.flat_grad = self.flat.flat_param, self.flat.flat_grad
        self.exp_avg = torch.zeros_like(self.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat_param)

    def zero_grad(self):
        self.flat.zero_grad()

    def step(self, lr=None):
        if lr is not None:
            self.lr = lr
        dev = self.flat_param.device
        if dev.type != "cuda":
            raise RuntimeError("FusedAdam.step: parameters must live on the GPU (HIP kernel only)")
        self.step_count += 1
        with torch.cuda.device(dev):
            st = _lib.lib().mpa_adam_step(
                _lib.ptr(self.flat_param), _lib.ptr(self.flat_grad), _lib.ptr(self.exp_avg),
                _lib.ptr(self.exp_avg_sq), self.numel, float(self.lr), float(self.betas[0]),
                float(self.betas[1]), float(self.eps), float(self.weight_decay), int(self.decoupled),
                self.step_count, float(self.grad_scale), _lib.current_stream(dev))
        _lib.check(st, "mpa_adam_step")

    # ---- graph-capturable form: hyper-parameters live in a 4-float device buffer -----------------------
    def _hyper_values(self, step):
        # same arithmetic as mpa_adam_step: betas rounded to fp32 first, powers and sqrt in double
        b1, b2 = (float(torch.tensor(b, dtype=torch.float32)) for b in self.betas)
        return [float(self.lr), 1.0 - b1 ** step, math.sqrt(1.0 - b2 ** step), float(self.grad_scale)]

    def prepare_hyper(self):
        """Advance the step count and upload {lr, bias corrections, grad_scale} for the next captured or
        eager `step_dev` launch (async copy from pinned memory on the current stream)."""
        if not hasattr(self, "_hyper_dev"):
            self._hyper_dev = torch.zeros(4, dtype=torch.float32, device=self.flat_param.device)
            self._hyper_host = torch.zeros(4, dtype=torch.float32).pin_memory()
        self.step_count += 1
        self._hyper_host.copy_(torch.tensor(self._hyper_values(self.step_count)))
        self._hyper_dev.copy_(self._hyper_host, non_blocking=True)

    def step_dev(self):
        """The Adam launch with device-resident hyper-parameters (safe to capture in a HIP graph)."""
        dev = self.flat_param.device
        with torch.cuda.device(dev):
            st = _lib.lib().mpa_adam_step_dev(
                _lib.ptr(self.flat_param), _lib.ptr(self.flat_grad), _lib.ptr(self.exp_avg),
                _lib.ptr(self.exp_avg_sq), self.numel, _lib.ptr(self._hyper_dev), float(self.betas[0]),
                float(self.betas[1]), float(self.eps), float(self.weight_decay), int(self.decoupled),
                _lib.current_stream(dev))
        _lib.check(st, "mpa_adam_step_dev")

    def state_dict(self):
        return {"step": self.step_count, "lr": self.lr, "exp_avg": self.exp_avg.clone(),
                "exp_avg_sq": self.exp_avg_sq.clone()}

    def load_state_dict(self, state):
        self.step_count, self.lr = state["step"], state["lr"]
        self.exp_avg.copy_(state["exp_avg"])
        self.exp_avg_sq.copy_(state["exp_avg_sq"])


def cosine_warmup_lr(total_epochs, warmup_epochs, max_lr, min_lr):
    """epoch -> learning rate.  Epoch 0 runs at `min_lr` (the reference scheduler resets the rate
    to min_lr at the end of its constructor, utils/lr.py:68-75), later epochs follow the linear
    warm-up / half-cosine of utils/lr.py:77-91 and restart every `total_epochs`."""
    assert warmup_epochs < total_epochs

    def lr_at(epoch):
        if epoch <= 0:
            return min_lr
        s = epoch % total_epochs
        if s < warmup_epochs:
            return (max_lr - min_lr) * s / warmup_epochs + min_lr
        phase = (s - warmup_epochs) / (total_epochs - warmup_epochs)
        return min_lr + (max_lr - min_lr) * (1 + math.cos(math.pi * phase)) / 2

    return lr_at
