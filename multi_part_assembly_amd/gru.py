"""Recurrent half of a (bi)directional GRU on the HIP library (csrc/gru.hip): one launch per pass for all steps and both
directions instead of the library's per-step launches.  The nn.GRU module keeps holding the parameters (same state_dict
keys); the input projections stay library GEMMs (autograd handles them)."""
from __future__ import annotations

import ctypes

import torch

from . import _lib


class _GRURecurrentFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gi, h0, whh, bhh):
        D, B, T, H3 = gi.shape
        H = H3 // 3
        dev = gi.device
        L = _lib.lib()
        n = ctypes.c_int64()
        _lib.check(L.mpa_gru_workspace(D, B, T, H, ctypes.byref(n)), "mpa_gru_workspace")
        ws = torch.empty(n.value, dtype=torch.float32, device=dev)
        out = torch.empty((D, B, T, H), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            tok = _lib.KernelTimer.start(f"gru_forward[{D}x{B}x{T}x{H}]")
            st = L.mpa_gru_forward(_lib.ptr(gi), _lib.ptr(h0), _lib.ptr(whh), _lib.ptr(bhh), D, B, T, H, _lib.ptr(ws),
                                   _lib.ptr(out), _lib.current_stream(dev))
            _lib.KernelTimer.stop(tok)
        _lib.check(st, "mpa_gru_forward")
        ctx.save_for_backward(h0, whh, out, ws)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        h0, whh, out, ws = ctx.saved_tensors
        D, B, T, H = out.shape
        dev = out.device
        ggi = torch.empty((D, B, T, 3 * H), dtype=torch.float32, device=dev)
        gw = torch.empty_like(whh)
        gb = torch.empty((D, 3 * H), dtype=torch.float32, device=dev)
        grad_out = grad_out.contiguous()
        with torch.cuda.device(dev):
            tok = _lib.KernelTimer.start(f"gru_backward[{D}x{B}x{T}x{H}]")
            st = _lib.lib().mpa_gru_backward(_lib.ptr(grad_out), _lib.ptr(h0), _lib.ptr(whh), _lib.ptr(out), D, B, T, H,
                                             _lib.ptr(ws), _lib.ptr(ggi), _lib.ptr(gw), _lib.ptr(gb),
                                             _lib.current_stream(dev))
            _lib.KernelTimer.stop(tok)
        _lib.check(st, "mpa_gru_backward")
        return ggi, None, gw, gb


_RESIDENT: dict = {}


def supported(hidden, batch, directions=2):
    """Shapes csrc/gru.hip is built for AND a device that can hold the kernels' whole grid at once (their per-step
    exchange needs every block resident: mpa_gru_resident asks the runtime's occupancy calculator; a partitioned or
    smaller device answers no and the caller keeps the library GRU)."""
    if not (hidden in (128, 256) and batch <= 64 and (48 * hidden + batch * hidden + batch * 64) * 4 <= 160 * 1024):
        return False
    key = (hidden, batch, directions, torch.cuda.current_device())
    if key not in _RESIDENT:
        ok = ctypes.c_int(0)
        _lib.check(_lib.lib().mpa_gru_resident(directions, batch, hidden, ctypes.byref(ok)), "mpa_gru_resident")
        _RESIDENT[key] = bool(ok.value)
    return _RESIDENT[key]


def gru_recurrent(gi, h0, whh, bhh):
    """gi [D, B, T, 3H] (input projections incl. b_ih, gate order r|z|n), h0 [D, B, H], whh [D, 3H, H], bhh [D, 3H]
    -> hidden states of every step [D, B, T, H] (torch.nn.GRU's equations)."""
    if not gi.is_cuda:
        raise RuntimeError("gru_recurrent: only CUDA (HIP) tensors are supported — no CPU fallback")
    return _GRURecurrentFn.apply(gi.float().contiguous(), h0.detach().float().contiguous(), whh.contiguous(),
                                 bhh.contiguous())
