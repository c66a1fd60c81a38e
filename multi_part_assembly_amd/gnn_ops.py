"""The small per-iteration pieces of the graph networks on the HIP library (csrc/gnn_glue.hip): the 7-wide first layer
of the pose encoder, the 512 -> 1 relation head with its sigmoid and valid-pair mask, the relation-weighted mean of the
edge features and the [part i ; part j] pair rows the edge MLP and the relation net read (reference models/dgl/modules.py:61-86, models/dgl/network.py:121-152).  The torch modules keep
holding the parameters (same state_dict keys); this only replaces what they compute."""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from .gradsink import GradSink


def _f32c(t):
    return t.to(torch.float32).contiguous()


class _NarrowLinearReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        R, K = x.shape
        N = weight.shape[0]
        out = torch.empty((R, N), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            tok = _lib.KernelTimer.start(f"narrow_linear_relu_forward[{R}x{K}x{N}]")
            st = _lib.lib().mpa_narrow_linear_relu_forward(_lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), R, K, N,
                                                           _lib.ptr(out), _lib.current_stream(x.device))
            _lib.KernelTimer.stop(tok)
        _lib.check(st, "mpa_narrow_linear_relu_forward")
        ctx.params = [p for p in (weight, bias) if p is not None]
        ctx.has_bias = bias is not None
        GradSink.note_use(ctx.params)
        ctx.save_for_backward(x, weight, out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, weight, out = ctx.saved_tensors
        R, K = x.shape
        N = weight.shape[0]
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        bufs, direct = GradSink.outputs(ctx.params)
        gw = bufs[0]
        gb = bufs[1] if ctx.has_bias else None
        grad_out = grad_out.contiguous()
        n = ctypes.c_int64()
        _lib.check(_lib.lib().mpa_narrow_linear_relu_workspace(R, K, N, ctypes.byref(n)), "mpa_narrow_linear_relu_workspace")
        ws = torch.empty(n.value, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            tok = _lib.KernelTimer.start(f"narrow_linear_relu_backward[{R}x{K}x{N}]")
            st = _lib.lib().mpa_narrow_linear_relu_backward(
                _lib.ptr(grad_out), _lib.ptr(out), _lib.ptr(x), _lib.ptr(weight), R, K, N, _lib.ptr(ws), _lib.ptr(gx),
                _lib.ptr(gw), _lib.ptr(gb), _lib.current_stream(x.device))
            _lib.KernelTimer.stop(tok)
        _lib.check(st, "mpa_narrow_linear_relu_backward")
        if direct:
            GradSink.delivered(ctx.params)
            return gx, None, None
        return gx, gw, gb


NARROW_MAX_IN = 16


def narrow_linear_relu(x, weight, bias=None):
    """relu(x W^T + b) for an input of at most 16 columns: x [..., K] -> [..., N]."""
    if not x.is_cuda:
        raise RuntimeError("narrow_linear_relu: only CUDA (HIP) tensors are supported — no CPU fallback")
    lead = x.shape[:-1]
    out = _NarrowLinearReLU.apply(_f32c(x).reshape(-1, x.shape[-1]), weight, bias)
    return out.view(*lead, weight.shape[0])


class _RelationHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, weight, bias, mask):
        R, K = h.shape
        dev = h.device
        n = ctypes.c_int64()
        _lib.check(_lib.lib().mpa_relation_head_workspace(R, K, ctypes.byref(n)), "mpa_relation_head_workspace")
        ws = torch.empty(n.value, dtype=torch.float32, device=dev)
        out = torch.empty(R, dtype=torch.float32, device=dev)
        w = weight.reshape(-1)  # the Linear weight [1, K]: the same bytes
        with torch.cuda.device(dev):
            tok = _lib.KernelTimer.start(f"relation_head_forward[{R}x{K}]")
            st = _lib.lib().mpa_relation_head_forward(_lib.ptr(h), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(mask), R, K,
                                                      _lib.ptr(ws), _lib.ptr(out), _lib.current_stream(dev))
            _lib.KernelTimer.stop(tok)
        _lib.check(st, "mpa_relation_head_forward")
        ctx.params = [p for p in (weight, bias) if p is not None]
        ctx.has_bias = bias is not None
        GradSink.note_use(ctx.params)
        ctx.save_for_backward(h, w, mask, ws)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        h, w, mask, ws = ctx.saved_tensors
        R, K = h.shape
        dev = h.device
        gh = torch.empty_like(h) if ctx.needs_input_grad[0] else None
        bufs, direct = GradSink.outputs(ctx.params)
        gw = bufs[0]
        gb = bufs[1] if ctx.has_bias else None
        grad_out = grad_out.contiguous()
        with torch.cuda.device(dev):
            tok = _lib.KernelTimer.start(f"relation_head_backward[{R}x{K}]")
            st = _lib.lib().mpa_relation_head_backward(
                _lib.ptr(grad_out), _lib.ptr(h), _lib.ptr(w), _lib.ptr(mask), R, K, _lib.ptr(ws), _lib.ptr(gh),
                _lib.ptr(gw), _lib.ptr(gb), _lib.current_stream(dev))
            _lib.KernelTimer.stop(tok)
        _lib.check(st, "mpa_relation_head_backward")
        if direct:
            GradSink.delivered(ctx.params)
            return gh, None, None, None
        return gh, gw, gb, None


def relation_head_supported(width):
    return width % 4 == 0 and 4 <= width <= 4096


def relation_head(h, weight, bias=None, mask=None):
    """sigmoid(h . w + b) [* mask]: h [..., K], weight [1, K] -> [...] (mask, if given, has the leading shape of h)."""
    if not h.is_cuda:
        raise RuntimeError("relation_head: only CUDA (HIP) tensors are supported — no CPU fallback")
    lead = h.shape[:-1]
    m = None if mask is None else _f32c(mask.detach()).reshape(-1)
    return _RelationHead.apply(_f32c(h).reshape(-1, h.shape[-1]), weight, bias, m).view(*lead)


class _RelationMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, edge, rel):
        G, P, C = edge.shape
        out = torch.empty((G, C), dtype=torch.float32, device=edge.device)
        with torch.cuda.device(edge.device):
            tok = _lib.KernelTimer.start(f"relation_mean_forward[{G}x{P}x{C}]")
            st = _lib.lib().mpa_relation_mean_forward(_lib.ptr(edge), _lib.ptr(rel), G, P, C, _lib.ptr(out),
                                                      _lib.current_stream(edge.device))
            _lib.KernelTimer.stop(tok)
        _lib.check(st, "mpa_relation_mean_forward")
        ctx.save_for_backward(edge, rel, out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        edge, rel, out = ctx.saved_tensors
        G, P, C = edge.shape
        ge = torch.empty_like(edge) if ctx.needs_input_grad[0] else None
        gr = torch.empty_like(rel) if ctx.needs_input_grad[1] else None
        grad_out = grad_out.contiguous()
        with torch.cuda.device(edge.device):
            tok = _lib.KernelTimer.start(f"relation_mean_backward[{G}x{P}x{C}]")
            st = _lib.lib().mpa_relation_mean_backward(_lib.ptr(grad_out), _lib.ptr(edge), _lib.ptr(rel), _lib.ptr(out), G,
                                                       P, C, _lib.ptr(ge), _lib.ptr(gr),
                                                       _lib.current_stream(edge.device))
            _lib.KernelTimer.stop(tok)
        _lib.check(st, "mpa_relation_mean_backward")
        return ge, gr


RELATION_MEAN_MAX_PARTS = 64


def relation_mean(edge, rel):
    """edge [B, P, P, C], rel [B, P, P] -> [B, P, C]: sum_j edge_ij rel_ij / (sum_j rel_ij + 1e-6)."""
    if not edge.is_cuda:
        raise RuntimeError("relation_mean: only CUDA (HIP) tensors are supported — no CPU fallback")
    B, P, P2, C = edge.shape
    out = _RelationMean.apply(_f32c(edge).reshape(B * P, P2, C), _f32c(rel).reshape(B * P, P2))
    return out.view(B, P, C)


class _PairRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, swap):
        S, P, F = a.shape
        out = torch.empty((S, P, P, 2 * F), dtype=torch.float32, device=a.device)
        with torch.cuda.device(a.device):
            tok = _lib.KernelTimer.start(f"pair_rows_forward[{S}x{P}x{F}]")
            st = _lib.lib().mpa_pair_rows_forward(_lib.ptr(a), _lib.ptr(b), S, P, F, int(swap), _lib.ptr(out),
                                                  _lib.current_stream(a.device))
            _lib.KernelTimer.stop(tok)
        _lib.check(st, "mpa_pair_rows_forward")
        ctx.dims = (S, P, F, int(swap))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        S, P, F, swap = ctx.dims
        dev = grad_out.device
        ga = torch.empty((S, P, F), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        gb = torch.empty((S, P, F), dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        grad_out = grad_out.contiguous()
        with torch.cuda.device(dev):
            tok = _lib.KernelTimer.start(f"pair_rows_backward[{S}x{P}x{F}]")
            st = _lib.lib().mpa_pair_rows_backward(_lib.ptr(grad_out), S, P, F, swap, _lib.ptr(ga), _lib.ptr(gb),
                                                   _lib.current_stream(dev))
            _lib.KernelTimer.stop(tok)
        _lib.check(st, "mpa_pair_rows_backward")
        return ga, gb, None


def pair_rows_supported(width):
    return width % 4 == 0 and 4 <= width <= 65536


def pair_rows(a, b, swap=False):
    """a, b [S, P, F] -> [S, P, P, 2F]: row (s, i, j) = [a[s, i] ; b[s, j]] (swap: [b[s, j] ; a[s, i]])."""
    if not a.is_cuda:
        raise RuntimeError("pair_rows: only CUDA (HIP) tensors are supported — no CPU fallback")
    return _PairRows.apply(_f32c(a), _f32c(b), bool(swap))
