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


def decay_mask_for(model, flat: "FlatBuffers"):
    """1/0 per element of the flat parameter buffer: the reference's AdamW parameter groups (utils/utils.py:90-125
    `filter_wd_parameters`) exempt every bias and the weights of normalisation layers from weight decay."""
    import torch.nn as nn
    norm_types = (nn.LayerNorm, nn.GroupNorm, nn.modules.batchnorm._BatchNorm, nn.modules.instancenorm._InstanceNorm)
    no_decay = set()
    for m in model.modules():
        if isinstance(m, norm_types) and getattr(m, "weight", None) is not None:
            no_decay.add(id(m.weight))
        if getattr(m, "bias", None) is not None and isinstance(m.bias, torch.Tensor):
            no_decay.add(id(m.bias))
    mask = torch.zeros_like(flat.flat_param)
    for p, off in zip(flat.params, flat.offsets):
        if id(p) not in no_decay:
            mask[off:off + p.numel()] = 1.0
    return mask


class FusedAdam:
    """One-kernel Adam / AdamW over FlatBuffers.  `step()` = `prepare_hyper()` + `step_dev()`: the step count and
    the bias corrections live in DEVICE memory and are advanced by the library on the stream (csrc/adam.hip), so
    neither eager steps nor HIP-graph replays upload per-step scalars from the host — a host that runs steps ahead
    of the GPU cannot corrupt a queued update.  Learning rate and grad_scale are written with `fill_` (the value
    travels as a kernel argument) only when they change.

    `decay_mask`: optional flat 1/0 tensor (see `decay_mask_for`).  `clip_grad`: global-norm clipping of the
    (mean) gradient, the reference's `gradient_clip_val` (scripts/train.py:90)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 decoupled_weight_decay=None, decay_mask=None, clip_grad=None):
        self.flat = params if isinstance(params, FlatBuffers) else FlatBuffers(params)
        self.params = self.flat.params
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        # the reference switches to AdamW as soon as weight_decay > 0 (base_model.py:394-404)
        self.decoupled = (weight_decay > 0) if decoupled_weight_decay is None else decoupled_weight_decay
        self.decay_mask = decay_mask
        self.clip_grad = clip_grad if clip_grad else None
        self._clip_ws = None
        self._clip_written = False  # the device-side clip coefficient holds a value other than 1
        self._step_count = 0
        self.grad_scale = 1.0
        self.numel = self.flat.numel
        self.flat_param, self.flat_grad = self.flat.flat_param, self.flat.flat_grad
        self.exp_avg = torch.zeros_like(self.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat_param)
        self._hyper_dev, self._uploaded = None, (None, None)

    def zero_grad(self):
        self.flat.zero_grad()

    # ---- device-resident hyper-parameters -----------------------------------------------------------------
    def _ensure_hyper(self):
        if self._hyper_dev is None:
            dev = self.flat_param.device
            if dev.type != "cuda":
                raise RuntimeError("FusedAdam.step: parameters must live on the GPU (HIP kernel only)")
            self._hyper_dev = torch.zeros(8, dtype=torch.float32, device=dev)
            self._hyper_dev[5] = 1.0  # clip coefficient: no clipping
            self._set_device_step(self._step_count)

    @property
    def step_count(self):
        return self._step_count

    @step_count.setter
    def step_count(self, value):
        """The device keeps its own copy (bias corrections are computed there): assigning re-synchronises it."""
        self._step_count = int(value)
        if self._hyper_dev is not None:
            self._set_device_step(self._step_count)

    def _ensure_clip_ws(self):
        """Allocated on first use, so `clip_grad` may be assigned after construction or after the first step."""
        if self._clip_ws is None:
            import ctypes
            nbytes = ctypes.c_int64()
            _lib.check(_lib.lib().mpa_grad_clip_workspace(ctypes.byref(nbytes)), "mpa_grad_clip_workspace")
            self._clip_ws = torch.empty(nbytes.value // 8, dtype=torch.float64, device=self.flat_param.device)

    def _set_device_step(self, step):
        self._hyper_dev[4:5].view(torch.int32).fill_(int(step))

    def prepare_hyper(self):
        """Host-side bookkeeping in front of `step_dev`: advances the host mirror of the step count and rewrites
        lr / grad_scale on the device when they changed (fill_: no host staging buffer involved)."""
        self._ensure_hyper()
        self._step_count += 1  # (the device advances its own counter inside step_dev)
        if self._uploaded != (float(self.lr), float(self.grad_scale)):
            self._hyper_dev[0:1].fill_(float(self.lr))
            self._hyper_dev[3:4].fill_(float(self.grad_scale))
            self._uploaded = (float(self.lr), float(self.grad_scale))

    def step_dev(self):
        """[clip coefficient +] step-count advance + Adam update, all on the stream (safe to capture in a HIP
        graph: every launch argument is constant across replays)."""
        dev = self.flat_param.device
        lib = _lib.lib()
        with torch.cuda.device(dev):
            stream = _lib.current_stream(dev)
            if not self.clip_grad and self._clip_written:  # clipping was switched off: back to "no clipping"
                self._hyper_dev[5:6].fill_(1.0)
                self._clip_written = False
            if self.clip_grad:
                self._ensure_clip_ws()
                self._clip_written = True
                st = lib.mpa_grad_clip_coef(_lib.ptr(self.flat_grad), self.numel, float(self.clip_grad),
                                            self._hyper_dev.data_ptr() + 12, 1.0, _lib.ptr(self._clip_ws),
                                            self._hyper_dev.data_ptr() + 20, stream)
                _lib.check(st, "mpa_grad_clip_coef")
            st = lib.mpa_adam_step_dev(
                _lib.ptr(self.flat_param), _lib.ptr(self.flat_grad), _lib.ptr(self.exp_avg),
                _lib.ptr(self.exp_avg_sq), self.numel, _lib.ptr(self._hyper_dev), float(self.betas[0]),
                float(self.betas[1]), float(self.eps), float(self.weight_decay), int(self.decoupled),
                _lib.ptr(self.decay_mask), stream)
        _lib.check(st, "mpa_adam_step_dev")

    def step(self, lr=None):
        if lr is not None:
            self.lr = lr
        self.prepare_hyper()
        self.step_dev()

    def state_dict(self):
        return {"step": self.step_count, "lr": self.lr, "exp_avg": self.exp_avg.clone(),
                "exp_avg_sq": self.exp_avg_sq.clone()}

    def load_state_dict(self, state):
        self.step_count, self.lr = state["step"], state["lr"]  # (the setter re-synchronises the device counter)
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
