"""Recurrent half of a (bi)directional GRU on the HIP library (csrc/gru.hip): one launch per pass for all steps and both
directions instead of the library's per-step launches.  The nn.GRU module keeps holding the parameters (same state_dict
keys); the input projections stay library GEMMs (autograd handles them)."""
from __future__ import annotations

import ctypes
import os

import torch

from . import _lib


# ---- the launches' status word ------------------------------------------------------------------------------------------
# The kernels' blocks wait for each other (csrc/gru.hip).  A launch whose blocks could not all be resident — another stream's
# kernels held the CUs the missing ones needed — gives up after its poll budget and raises ONE int32 in device memory instead
# of trapping.  That word lives here, per device, for the life of the process; every launch is followed by an asynchronous
# copy of it into pinned host memory (a graph node under capture), and the host looks at the pinned copy — a plain memory
# read, no synchronisation — in front of every later GRU call and in `raise_if_failed()`.
_STATUS: dict = {}


def _status(dev):
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _STATUS.get(idx)
    if st is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("gru: the status word must be allocated before a HIP-graph capture (gru.prepare(device), "
                               "done by Trainer.__init__, or one eager warm-up step)")
        dev = torch.device("cuda", idx)
        st = _STATUS[idx] = (torch.zeros(1, dtype=torch.int32, device=dev),
                                   torch.zeros(1, dtype=torch.int32).pin_memory())
    return st


def prepare(dev):
    """Allocate the device's status word and its pinned copy now (Trainer.__init__ calls this): both must exist before a
    HIP-graph capture reaches the first GRU launch."""
    return _status(torch.device(dev) if not isinstance(dev, torch.device) else dev)


def raise_if_failed(device=None, synchronize=False):
    """RuntimeError if a GRU launch on `device` (default: every device used so far) gave up because its blocks were not
    co-resident.  The outputs of such a launch are undefined.  `synchronize=True` first waits for the device, so that
    the answer covers every launch issued so far (otherwise: every launch whose status copy has already arrived)."""
    for idx, (word, host) in list(_STATUS.items()):
        if device is not None and torch.device(device).index not in (None, idx):
            continue
        if synchronize:
            torch.cuda.synchronize(idx)
        if int(host[0]) != 0:
            word.zero_()
            host.zero_()
            raise RuntimeError(
                "gru: a launch of csrc/gru.hip gave up — its blocks exchange one word per step and must all be resident at "
                "once, but some were never dispatched while the others waited (kernels of another stream holding the "
                "compute units?).  The outputs of that step are undefined; rerun it without the competing work, or set "
                "MPA_GRU=library to keep the library GRU")


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
        raise_if_failed(dev)
        word, host = _status(dev)
        with torch.cuda.device(dev):
            tok = _lib.KernelTimer.start(f"gru_forward[{D}x{B}x{T}x{H}]")
            st = L.mpa_gru_forward(_lib.ptr(gi), _lib.ptr(h0), _lib.ptr(whh), _lib.ptr(bhh), D, B, T, H, _lib.ptr(ws),
                                   _lib.ptr(out), _lib.ptr(word), _lib.current_stream(dev))
            _lib.KernelTimer.stop(tok)
            host.copy_(word, non_blocking=True)
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
        raise_if_failed(dev)
        word, host = _status(dev)
        with torch.cuda.device(dev):
            tok = _lib.KernelTimer.start(f"gru_backward[{D}x{B}x{T}x{H}]")
            st = _lib.lib().mpa_gru_backward(_lib.ptr(grad_out), _lib.ptr(h0), _lib.ptr(whh), _lib.ptr(out), D, B, T, H,
                                             _lib.ptr(ws), _lib.ptr(ggi), _lib.ptr(gw), _lib.ptr(gb), _lib.ptr(word),
                                             _lib.current_stream(dev))
            _lib.KernelTimer.stop(tok)
            host.copy_(word, non_blocking=True)
        _lib.check(st, "mpa_gru_backward")
        return ggi, None, gw, gb


_RESIDENT: dict = {}


def supported(hidden, batch, directions=2):
    """Shapes csrc/gru.hip is built for AND a device that can hold the kernels' whole grid at once (their per-step
    exchange needs every block resident: mpa_gru_resident asks the runtime's occupancy calculator; a partitioned or
    smaller device answers no and the caller keeps the library GRU)."""
    if os.environ.get("MPA_GRU", "") == "library":  # the way out when other streams' kernels keep the grid from being
        return False                                 # co-resident (see raise_if_failed)
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
