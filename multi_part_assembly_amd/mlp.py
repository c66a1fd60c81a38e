"""One MLP layer (Linear / Conv1d k=1 [+ BatchNorm1d] [+ ReLU]) over rows on the HIP library (csrc/mlp.hip) — the building
block of the graph networks' edge / node MLPs and relation nets (reference models/dgl/modules.py:5-73,
models/rgl_net/modules.py:5-30).  The torch modules keep holding the parameters (same state_dict keys); this only
replaces what `conv(x)`, `bn(.)`, `relu(.)` compute."""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from .gradsink import GradSink


class _MLPLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running, training, momentum, eps, relu):
        R, K = x.shape
        N = weight.shape[0]
        w = weight.reshape(N, -1)  # a Conv1d weight [N, K, 1] or a Linear weight [N, K]: the same bytes
        dev = x.device
        L = _lib.lib()
        nbytes = ctypes.c_int64()
        _lib.check(L.mpa_mlp_layer_workspace(R, K, N, ctypes.byref(nbytes)), "mpa_mlp_layer_workspace")
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
        out = torch.empty((R, N), dtype=torch.float32, device=dev)
        rm, rv = running if running is not None else (None, None)
        with torch.cuda.device(dev):
            tok = _lib.KernelTimer.start(f"mlp_layer_forward[{R}x{K}x{N}]")
            st = L.mpa_mlp_layer_forward(_lib.ptr(x), x.stride(0), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(gamma),
                                         _lib.ptr(beta), _lib.ptr(rm), _lib.ptr(rv), int(training), float(momentum),
                                         float(eps), int(relu), R, K, N, _lib.ptr(ws), _lib.ptr(out),
                                         _lib.current_stream(dev))
            _lib.KernelTimer.stop(tok)
        _lib.check(st, "mpa_mlp_layer_forward")
        ctx.meta = (bool(relu), bool(training), bias is not None, gamma is not None)
        # the parameters themselves (not views of them): with a GradSink active their gradients are written straight
        # into the flat gradient buffer instead of going through one AccumulateGrad `add` launch per parameter
        ctx.params = [p for p in (weight, bias, gamma, beta) if p is not None]
        GradSink.note_use(ctx.params)
        ctx.save_for_backward(x, w, gamma, out, ws)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        if getattr(ctx, "consumed", False):
            raise RuntimeError("MLP layer: the backward pass overwrites its saved workspace: a second backward over the "
                               "same forward (retain_graph=True) is not supported — run the forward again")
        ctx.consumed = True
        x, w, gamma, out, ws = ctx.saved_tensors
        relu, training, has_bias, has_bn = ctx.meta
        if has_bn and not training:
            raise RuntimeError("MLP layer: backward is implemented for training-mode BatchNorm only")
        R, K = x.shape
        N = w.shape[0]
        dev = x.device
        gx = torch.empty((R, K), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        bufs, direct = GradSink.outputs(ctx.params)
        it = iter(bufs)
        gw = next(it)
        gb = next(it) if has_bias else None
        gg = next(it) if has_bn else None
        gbe = next(it) if has_bn else None
        grad_out = grad_out.contiguous()
        with torch.cuda.device(dev):
            tok = _lib.KernelTimer.start(f"mlp_layer_backward[{R}x{K}x{N}]")
            st = _lib.lib().mpa_mlp_layer_backward(
                _lib.ptr(grad_out), _lib.ptr(x), x.stride(0), _lib.ptr(w), _lib.ptr(gamma), _lib.ptr(out), int(relu), R, K,
                N, _lib.ptr(ws), _lib.ptr(gx), _lib.ptr(gw), _lib.ptr(gb), _lib.ptr(gg), _lib.ptr(gbe),
                _lib.current_stream(dev))
            _lib.KernelTimer.stop(tok)
        _lib.check(st, "mpa_mlp_layer_backward")
        if direct:
            GradSink.delivered(ctx.params)
            return gx, None, None, None, None, None, None, None, None, None
        return gx, gw, gb, gg, gbe, None, None, None, None, None


class _PairLayerFn(torch.autograd.Function):
    """First layer of the P x P edge MLP on (a_i, b_j) pairs without the pair tensor (csrc/mlp.hip: mpa_pair_layer_*)."""

    @staticmethod
    def forward(ctx, a, b, weight, bias, gamma, beta, running, training, momentum, eps, relu):
        B, P, F = a.shape
        N = weight.shape[0]
        w = weight.reshape(N, -1)  # Conv1d weight [N, 2F, 1]: the same bytes
        dev = a.device
        L = _lib.lib()
        nbytes = ctypes.c_int64()
        _lib.check(L.mpa_pair_layer_workspace(B, P, F, N, ctypes.byref(nbytes)), "mpa_pair_layer_workspace")
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
        out = torch.empty((B * P * P, N), dtype=torch.float32, device=dev)
        rm, rv = running
        with torch.cuda.device(dev):
            tok = _lib.KernelTimer.start(f"pair_layer_forward[{B}x{P}x{P}x{2 * F}x{N}]")
            st = L.mpa_pair_layer_forward(_lib.ptr(a), _lib.ptr(b), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(gamma),
                                          _lib.ptr(beta), _lib.ptr(rm), _lib.ptr(rv), int(training), float(momentum),
                                          float(eps), int(relu), B, P, F, N, _lib.ptr(ws), _lib.ptr(out),
                                          _lib.current_stream(dev))
            _lib.KernelTimer.stop(tok)
        _lib.check(st, "mpa_pair_layer_forward")
        ctx.meta = (bool(relu), bool(training), bias is not None)
        ctx.same = a.data_ptr() == b.data_ptr() and a.shape == b.shape
        ctx.params = [p for p in (weight, bias, gamma, beta) if p is not None]
        GradSink.note_use(ctx.params)
        ctx.save_for_backward(a, b, w, gamma, out, ws)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        if getattr(ctx, "consumed", False):
            raise RuntimeError("pair layer: the backward pass overwrites its saved workspace: a second backward over the "
                               "same forward (retain_graph=True) is not supported — run the forward again")
        ctx.consumed = True
        a, b, w, gamma, out, ws = ctx.saved_tensors
        relu, training, has_bias = ctx.meta
        if not training:
            raise RuntimeError("pair layer: backward is implemented for training-mode BatchNorm only")
        B, P, F = a.shape
        N = w.shape[0]
        dev = a.device
        ga = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        gb = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        same = ctx.same and ga is not None and gb is not None
        if same:
            gb = ga  # one tensor in both roles: the library adds the second half's gradient into the first's buffer
        bufs, direct = GradSink.outputs(ctx.params)
        it = iter(bufs)
        gw = next(it)
        gbias = next(it) if has_bias else None
        gg, gbe = next(it), next(it)
        grad_out = grad_out.contiguous()
        with torch.cuda.device(dev):
            tok = _lib.KernelTimer.start(f"pair_layer_backward[{B}x{P}x{P}x{2 * F}x{N}]")
            st = _lib.lib().mpa_pair_layer_backward(
                _lib.ptr(grad_out), _lib.ptr(a), _lib.ptr(b), _lib.ptr(w), _lib.ptr(gamma), _lib.ptr(out), int(relu), B, P, F,
                N, _lib.ptr(ws), _lib.ptr(ga), _lib.ptr(gb), _lib.ptr(gw), _lib.ptr(gbias), _lib.ptr(gg), _lib.ptr(gbe),
                _lib.current_stream(dev))
            _lib.KernelTimer.stop(tok)
        _lib.check(st, "mpa_pair_layer_backward")
        if same:
            gb = None  # (already inside ga)
        if direct:
            GradSink.delivered(ctx.params)
            return (ga, gb) + (None,) * 9
        return ga, gb, gw, gbias, gg, gbe, None, None, None, None, None


def pair_layer_supported(feat, out_dim):
    return feat % 64 == 0 and 64 <= feat <= 2048 and supported(2 * feat, out_dim)


def pair_layer(a, b, weight, bias, bn, relu=True, training=True):
    """a, b [B, P, F] (CUDA) -> [B*P*P, N] = act(bn([a_i ; b_j] W^T + bias)), row (s, i, j) at (s*P + i)*P + j: the first
    Conv1d(k=1) + BatchNorm1d (+ ReLU) of the edge MLP over all part pairs, computed from two GEMMs over the B*P part rows
    (the pair tensor of dgl/network.py:135-144 is never built).  `weight` [N, 2F] or the Conv1d weight [N, 2F, 1]."""
    if not a.is_cuda:
        raise RuntimeError("pair_layer: only CUDA (HIP) tensors are supported — no CPU fallback")
    a, b = a.float().contiguous(), b.float().contiguous()
    if training:
        _count_batch(bn)
    return _PairLayerFn.apply(a, b, weight, bias, bn.weight, bn.bias, (bn.running_mean, bn.running_var), training,
                              bn.momentum, bn.eps, relu)


class bn_counter_batch:
    """`with bn_counter_batch():` — the `num_batches_tracked += 1` of every BatchNorm layer that `mlp_layer` runs in
    training mode inside the block becomes ONE multi-tensor launch at its end instead of one launch per layer (18 per
    step for the graph networks).  A layer used k times in the block is advanced by k."""
    pending = None

    def __enter__(self):
        self.prev, bn_counter_batch.pending = bn_counter_batch.pending, {}
        return self

    def __exit__(self, *exc):
        items, bn_counter_batch.pending = bn_counter_batch.pending, self.prev
        if items:
            with torch.no_grad():
                torch._foreach_add_([t for t, _ in items.values()], [k for _, k in items.values()])
        return False


def _count_batch(bn):
    """bn.num_batches_tracked += 1 — now, or with the other layers' counters at the end of a bn_counter_batch block."""
    pend, t = bn_counter_batch.pending, bn.num_batches_tracked
    if pend is not None:
        pend[id(t)] = (t, pend.get(id(t), (t, 0))[1] + 1)
    else:
        with torch.no_grad():
            t += 1


def supported(in_dim, out_dim):
    return in_dim % 64 == 0 and out_dim % 64 == 0 and 64 <= in_dim <= 4096 and 64 <= out_dim <= 4096


def mlp_layer(x, weight, bias=None, bn=None, relu=True, training=True):
    """x [R, K] (CUDA, fp32, rows contiguous) -> [R, N] = act(bn(x W^T + b)).  `weight` [N, K] or a Conv1d weight
    [N, K, 1]; `bn`: an nn.BatchNorm1d (its running statistics are updated in training mode) or None."""
    if not x.is_cuda:
        raise RuntimeError("mlp_layer: only CUDA (HIP) tensors are supported — no CPU fallback")
    x = x.float()
    if x.stride(1) != 1 or x.stride(0) % 4 != 0:
        x = x.contiguous()
    if bn is None:
        return _MLPLayerFn.apply(x, weight, bias, None, None, None, training, 0.0, 0.0, relu)
    if training:
        _count_batch(bn)
    return _MLPLayerFn.apply(x, weight, bias, bn.weight, bn.bias, (bn.running_mean, bn.running_var), training,
                             bn.momentum, bn.eps, relu)
