"""Direct gradient delivery for the HIP autograd Functions.

By default a custom Function returns freshly allocated gradient tensors and autograd's AccumulateGrad adds
each of them into `param.grad` — one tiny `add` launch per parameter (81 per step for PNTransformer, ~0.3 ms).
When a `GradSink` is active (the Trainer activates one around forward + backward) the backward kernels write
straight into `param.grad` — which `optim.FlatBuffers` made a view of the flat gradient buffer that the fused
Adam step and the data-parallel all-reduce operate on — and the Function returns None for those inputs.

Safety rule: overwriting is only equivalent to accumulating when the parameter receives exactly ONE gradient
contribution in the step and its .grad was zeroed before.  The sink therefore counts, per step, how many Function
calls used each parameter (`note_use`, called from forward); a parameter used more than once (e.g. the pose
head under MoN sampling, base_model.py:113-148 upstream) falls back to the returned-tensor path.  `on_ready` is
the data-parallel bucket hook (dp.BucketedGradReducer._on_grad), which AccumulateGrad would otherwise trigger.
"""
from __future__ import annotations

import torch


class GradSink:
    active: "GradSink | None" = None

    def __init__(self, on_ready=None):
        self.uses: dict[int, int] = {}
        self.on_ready = on_ready

    def __enter__(self):
        self.uses.clear()
        GradSink.active = self
        return self

    def __exit__(self, *exc):
        GradSink.active = None
        return False

    # ---- called by the autograd Functions ------------------------------------------------------------------
    @staticmethod
    def note_use(params):
        sink = GradSink.active
        if sink is not None:
            for p in params:
                sink.uses[id(p)] = sink.uses.get(id(p), 0) + 1

    @staticmethod
    def outputs(params):
        """-> (buffers, direct): the tensors the backward kernels must overwrite with d loss / d param."""
        sink = GradSink.active
        if sink is not None and all(
                sink.uses.get(id(p)) == 1 and p.is_leaf and p.grad is not None and p.grad.is_contiguous()
                and p.grad.dtype == torch.float32 and p.grad.device == p.device for p in params):
            return [p.grad for p in params], True
        return [torch.empty_like(p) for p in params], False

    @staticmethod
    def delivered(params):
        sink = GradSink.active
        if sink is not None and sink.on_ready is not None:
            for p in params:
                sink.on_ready(p)


class DeferredBackward:
    """Lets the Trainer cut the backward pass in front of the part encoder (graph-mode data parallelism).

    While `DeferredBackward.active` is a list, an encoder Function whose only gradients are parameter gradients (its
    input points are data) does not run its backward kernels when autograd reaches it: it parks (run, grad_output) here
    and returns.  `run_all()` then executes the parked backward calls — the Trainer captures that into a SECOND HIP graph,
    so the all-reduce of the first gradient bucket (everything but the encoder) can be launched between the two graphs
    and overlaps the encoder's backward, exactly like the eager path's bucket hooks."""
    active: "list | None" = None

    @staticmethod
    def park(run, grad, params=()):
        """`params`: the parameters `run` writes gradients of — the Trainer checks that all of them live in the LAST
        gradient bucket before it overlaps the first bucket's all-reduce with the parked calls."""
        DeferredBackward.active.append((run, grad, tuple(params)))

    @staticmethod
    def run_all():
        items, DeferredBackward.active = DeferredBackward.active or [], None
        for run, grad, _ in items:
            run(grad)
        return len(items)
