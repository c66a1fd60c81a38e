"""Minimal training harness standing in for `pl.Trainer` on the hot path (the reference drives
`BaseModel.training_step` through Lightning's automatic-optimisation loop, scripts/train.py:82-120):
zero_grad -> training_step -> backward (gradient buckets all-reduced while it runs) -> fused Adam,
with the per-epoch cosine schedule.  No host synchronisation inside a step: the loss is returned as
a device tensor.

With `use_graph=True` the whole step (several hundred launches, most of them tiny) is captured once
into a HIP graph and replayed: the launch shapes are static by construction (all B*P part slots +
masks, no compaction), the batch is copied into static input tensors, and the optimiser's
per-step scalars live in device memory.  On one GPU the graph holds zero_grad + forward + backward +
Adam.  With data parallelism the step is TWO graphs cut in front of the part encoder's backward
(`DeferredBackward`): graph A = zero_grad + forward + backward of everything but the encoder, then the
all-reduce of the first gradient bucket is launched (asynchronously, on the collective's stream), graph B =
the encoder's backward runs under it, then the second bucket's all-reduce and Adam — the same two-bucket
overlap as the eager path's hooks.

Status: bit-identical to eager launches (tests/test_model_gpu.py) and stable at the full benchmark size.  Two
things make that true: the library issues no hipMemsetAsync / hipMemcpyAsync (as graph NODES they returned stale
data on replay — what an earlier revision of this note blamed on a library GEMM), and everything that must differ
between replays is read from device memory that the host refreshes before each replay: the optimiser's step
scalars (csrc/adam.hip keeps the step count on the device) and the dropout seeds (`advance_seed`).  The step is
GPU-bound, so the replay is only ~2 % faster than eager launches; eager stays the default of bench.py.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import gru as _gru
from .gradsink import DeferredBackward, GradSink
from .dp import BucketedGradReducer, broadcast_from_rank0, bucket_sizes_for, ordered_parameters
from .optim import FlatBuffers, FusedAdam, cosine_warmup_lr, decay_mask_for


class Trainer:
    def __init__(self, model, cfg=None, process_group=None, use_graph=False, graph_warmup=3):
        self.model = model
        if use_graph and getattr(model, "semantic", False):
            # identical-part matching draws its point sample on the host every step (base_model._match_parts; reference
            # base_model.py:196-238): a captured step would replay ONE draw for ever, and the capture itself would hit the
            # host copy of the match ids.  Such models keep their launches eager.
            import warnings
            warnings.warn("Trainer: use_graph=True is not available for models with semantic part matching (the matching's "
                          "point sample is drawn on the host every step); running eager launches")
            use_graph = False
        self.use_graph, self.graph_warmup = use_graph, graph_warmup
        self._graph, self._static_batch, self._static_loss, self._eager_steps = None, None, None, 0
        cfg = cfg if cfg is not None else model.cfg
        opt = cfg.optimizer
        self.flat = FlatBuffers(ordered_parameters(model))
        # weight decay > 0 -> AdamW with biases / normalisation weights exempt (base_model.py:394-404); clip_grad ->
        # global-norm clipping of the averaged gradient (scripts/train.py:90 gradient_clip_val)
        self.optimizer = FusedAdam(self.flat, lr=opt.lr, weight_decay=opt.weight_decay,
                                   decay_mask=decay_mask_for(model, self.flat) if opt.weight_decay > 0 else None,
                                   clip_grad=opt.get("clip_grad", None))
        self.schedule = None
        if opt.lr_scheduler:
            total = cfg.exp.num_epochs
            self.schedule = cosine_warmup_lr(total, int(total * opt.warmup_ratio), opt.lr,
                                             opt.lr / opt.lr_decay_factor)
        self.epoch = 0
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        broadcast_from_rank0(self.flat, model, process_group)
        self.group = process_group
        # eager: bucket hooks fire during backward; graph mode: the same two buckets, reduced between / behind the two
        # graphs (the reducer only provides the bucket slices there: its hooks are switched off)
        self.reducer = BucketedGradReducer(self.flat, bucket_sizes_for(model, self.flat), process_group)
        if use_graph:
            self.reducer.enabled = False
        self._graph_b = None
        self.sink = GradSink(on_ready=self.reducer._on_grad if not use_graph and self.world > 1 else None)
        # buffers the step allocates lazily (the GRU launches' status word + its pinned copy, the loss's side stream)
        # exist BEFORE a capture can see them: a pinned allocation is illegal under capture and a device word born there
        # would live in the graph's private pool
        dev = self.flat.flat_param.device
        if dev.type == "cuda":
            _gru.prepare(dev)
            for mod in model.modules():
                if hasattr(mod, "prepare_streams"):
                    mod.prepare_streams(dev)
        self.set_epoch(0)

    def check_health(self, synchronize=False):
        """RuntimeError if a cooperative launch of this step (csrc/gru.hip) gave up: its outputs, and every gradient
        behind them, are undefined.  Without `synchronize` the answer covers every launch whose status copy has already
        arrived (a plain read of pinned memory: this runs after every step and replay); with it — in front of
        checkpoints and evaluation reports — every launch issued so far."""
        _gru.raise_if_failed(self.flat.flat_param.device, synchronize=synchronize)

    def state_dict(self):
        """Model + optimiser state for a checkpoint; refuses (RuntimeError) if a launch behind the current parameters
        gave up."""
        self.check_health(synchronize=True)
        return {"model": self.model.state_dict(), "optimizer": self.optimizer.state_dict()
                if hasattr(self.optimizer, "state_dict") else None, "epoch": self.epoch}

    def set_epoch(self, epoch):
        self.epoch = epoch
        if self.schedule is not None:
            self.optimizer.lr = self.schedule(epoch)

    # ---- graph mode -----------------------------------------------------------------------------------
    def _tensor_items(self, data_dict):
        return {k: v for k, v in data_dict.items() if isinstance(v, torch.Tensor)}

    def _fwd_bwd(self, batch, loss_out=None):
        self.optimizer.zero_grad()
        with self.sink:  # HIP backward kernels write straight into the flat gradient buffer
            loss = self.model.training_step(batch, 0)
            if loss_out is not None:  # graph mode: the loss is copied out before backward recycles memory
                loss_out.copy_(loss.detach())
            loss.backward()
        return loss.detach() if loss_out is None else loss_out

    def _capture(self, data_dict):
        self._static_batch = {k: v.clone() for k, v in self._tensor_items(data_dict).items()}
        self._loss_buf = torch.zeros((), dtype=torch.float32, device=self.flat.flat_param.device)
        self._graph = torch.cuda.CUDAGraph()
        if self.world == 1:
            # thread_local: RCCL's watchdog thread may touch the HIP runtime while we capture
            with torch.cuda.graph(self._graph, capture_error_mode="thread_local"):
                self._static_loss = self._fwd_bwd(self._static_batch, self._loss_buf)
                self.optimizer.step_dev()
            return
        # data parallel: graph A stops in front of the encoder's backward, graph B is the encoder's backward
        self.sink.__enter__()  # one sink (one set of use counts) across both graphs
        try:
            with torch.cuda.graph(self._graph, capture_error_mode="thread_local"):
                self.optimizer.zero_grad()
                DeferredBackward.active = []
                loss = self.model.training_step(self._static_batch, 0)
                self._loss_buf.copy_(loss.detach())
                loss.backward()
            self._static_loss = self._loss_buf
            parked = DeferredBackward.active
            if parked:
                self._check_parked_in_last_bucket(parked)
                self._graph_b = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._graph_b, pool=self._graph.pool(), capture_error_mode="thread_local"):
                    DeferredBackward.run_all()
        finally:
            DeferredBackward.active = None
            self.sink.__exit__(None, None, None)

    def _check_parked_in_last_bucket(self, parked):
        """The two-graph overlap reduces bucket 0 WHILE the parked backward calls run: every gradient they write must
        lie in the last bucket (an encoder module living outside `model.encoder` would race with the all-reduce)."""
        buckets = self.reducer.buckets
        if len(buckets) < 2:
            return
        if len(buckets) > 2:  # (checked here, before the second graph is captured — not at the first replay)
            raise RuntimeError(f"graph-mode data parallelism overlaps exactly two gradient buckets, the reducer made "
                               f"{len(buckets)}; use use_graph=False for this model")
        last = buckets[-1]["slice"]
        base = self.flat.flat_grad.data_ptr()
        lo = (last.data_ptr() - base) // last.element_size()
        hi = lo + last.numel()
        # by the parameters' SLOTS in the flat gradient buffer, not by p.grad: a parameter whose .grad is still None (or
        # that autograd accumulates elsewhere) must not pass unchecked
        slot = {id(p): off for p, off in zip(self.flat.params, self.flat.offsets)}
        for _, _, params in parked:
            for p in params:
                if not isinstance(p, torch.Tensor):
                    continue
                off = slot.get(id(p))
                if off is None or not (lo <= off < hi):
                    raise RuntimeError("graph-mode data parallelism: a deferred encoder backward owns a parameter "
                                       "outside the last gradient bucket (an encoder module outside model.encoder?); "
                                       "use use_graph=False for this model")

    def _reduce_buckets_around(self, run_second_half):
        """bucket 0 (everything but the encoder) is reduced WHILE `run_second_half` (the encoder's backward) runs; the
        encoder's bucket behind it."""
        buckets = self.reducer.buckets
        assert len(buckets) <= 2, "the two-graph overlap reduces only the first and the last bucket"
        first = dist.all_reduce(buckets[0]["slice"], group=self.group, async_op=True) if len(buckets) > 1 else None
        run_second_half()
        last = dist.all_reduce(buckets[-1]["slice"], group=self.group, async_op=True)
        if first is not None:
            first.wait()
        last.wait()

    def _graph_step(self, data_dict):
        self.model.train()
        self.optimizer.grad_scale = 1.0 / self.world
        if self._graph is None:
            if self._eager_steps < self.graph_warmup:  # let allocator / library state settle first
                self._eager_steps += 1
                loss = self._fwd_bwd(self._tensor_items(data_dict))
                if self.world > 1:
                    dist.all_reduce(self.flat.flat_grad, group=self.group)
                self.check_health()
                self.optimizer.prepare_hyper()
                self.optimizer.step_dev()
                return loss
            torch.cuda.synchronize()
            self._capture(data_dict)
        for k, v in self._tensor_items(data_dict).items():
            if v.data_ptr() != self._static_batch[k].data_ptr():
                self._static_batch[k].copy_(v, non_blocking=True)
        self.optimizer.prepare_hyper()
        for mod in self.model.modules():  # fresh dropout masks: the captured kernels read their seed from memory
            if hasattr(mod, "advance_seed"):
                mod.advance_seed()
        self._graph.replay()
        if self.world > 1:
            if self._graph_b is not None:
                self._reduce_buckets_around(self._graph_b.replay)
            else:  # nothing was deferred (an encoder outside the fused kernels): one all-reduce behind the replay
                dist.all_reduce(self.flat.flat_grad, group=self.group)
            self.check_health()  # before the (uncaptured) optimiser step consumes the gradients
            self.optimizer.step_dev()
        # the captured status copy refreshes the pinned word on every replay: a replay that gave up is reported here, at
        # the latest one replay later (the copy is asynchronous), and the word is cleared so that later replays run
        self.check_health()
        return self._static_loss

    def train_step(self, data_dict, batch_idx=0):
        """One optimiser step on this rank's shard; returns the (detached) scalar loss tensor."""
        if self.use_graph:
            return self._graph_step(data_dict)
        self.model.train()
        self.optimizer.zero_grad()
        with self.sink:  # HIP backward kernels write straight into the flat gradient buffer
            loss = self.model.training_step(data_dict, batch_idx)
            one = getattr(self, "_one", None)  # d loss / d loss, kept: autograd would fill a fresh ones_like every step
            if one is None or one.device != loss.device or one.dtype != loss.dtype:
                one = self._one = torch.ones((), dtype=loss.dtype, device=loss.device)
            loss.backward(one)
        self.optimizer.grad_scale = self.reducer.finish()
        self.check_health()  # a launch that gave up must not reach the parameters (csrc/gru.hip's status word)
        self.optimizer.step()
        return loss.detach()
