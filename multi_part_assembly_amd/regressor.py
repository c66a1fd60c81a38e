"""Pose head — mirror of the reference's PoseRegressor / StocasticPoseRegressor
(multi_part_assembly/models/modules/regressor.py:30-84); identical state_dict keys
(`fc_layers.{0,2}.*`, `rot_head.*`, `trans_head.*`).  Quaternion output only.

Compute: csrc/transformer.hip (`mpa_pose_head_*`): two fp32-MFMA GEMMs with the LeakyReLU fused, one
kernel for both heads + the quaternion normalisation, deterministic backward.
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .gradsink import GradSink


class _PoseHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, *params):
        M, Fdim = x.shape
        dev = x.device
        lib = _lib.lib()
        n = ctypes.c_int64()
        _lib.check(lib.mpa_pose_head_workspace(M, Fdim, ctypes.byref(n)), "mpa_pose_head_workspace")
        ws = torch.empty(n.value, dtype=torch.float32, device=dev)
        rot = torch.empty((M, 4), dtype=torch.float32, device=dev)
        trans = torch.empty((M, 3), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            tok = _lib.KernelTimer.start(f"pose_head_forward[{M}x{Fdim}]")
            st = lib.mpa_pose_head_forward(_lib.ptr(x), _lib.ptr_array(params), M, Fdim, _lib.ptr(ws),
                                           _lib.ptr(rot), _lib.ptr(trans), _lib.current_stream(dev))
            _lib.KernelTimer.stop(tok)
        _lib.check(st, "mpa_pose_head_forward")
        ctx.params = params
        GradSink.note_use(params)
        ctx.save_for_backward(x, ws)
        return rot, trans

    @staticmethod
    def backward(ctx, grad_rot, grad_trans):
        x, ws = ctx.saved_tensors
        params = ctx.params
        M, Fdim = x.shape
        dev = x.device
        grad_x = torch.empty_like(x)
        grads, direct = GradSink.outputs(params)
        with torch.cuda.device(dev):
            tok = _lib.KernelTimer.start(f"pose_head_backward[{M}x{Fdim}]")
            st = _lib.lib().mpa_pose_head_backward(
                _lib.ptr(grad_rot.contiguous()), _lib.ptr(grad_trans.contiguous()), _lib.ptr(x),
                _lib.ptr_array(params), M, Fdim, _lib.ptr(ws), _lib.ptr(grad_x), _lib.ptr_array(grads),
                _lib.current_stream(dev))
            _lib.KernelTimer.stop(tok)
        _lib.check(st, "mpa_pose_head_backward")
        if direct:
            GradSink.delivered(params)
            return (grad_x, *([None] * len(params)))
        return (grad_x, *grads)


class PoseRegressor(nn.Module):
    def __init__(self, feat_dim, rot_type="quat", norm_rot=True):
        super().__init__()
        if rot_type != "quat":
            raise NotImplementedError(f"rotation {rot_type} is not supported")
        self.rot_type, self.norm_rot = rot_type, norm_rot
        self.fc_layers = nn.Sequential(nn.Linear(feat_dim, 256), nn.LeakyReLU(0.2),
                                       nn.Linear(256, 128), nn.LeakyReLU(0.2))
        self.rot_head = nn.Linear(128, 4)
        self.trans_head = nn.Linear(128, 3)
        self.native = norm_rot  # csrc/transformer.hip normalises the quaternion in its head kernel

    def forward(self, x):
        """x [B, C] or [B, P, C] -> (rot [.., 4] unit-normalised, trans [.., 3])."""
        if not x.is_cuda:
            raise RuntimeError("PoseRegressor: only CUDA (HIP) tensors are supported — no CPU fallback")
        if self.native:
            lead = x.shape[:-1]
            # (any input width: the library zero-pads odd widths — labels / noise appended — to its 64-column GEMM panels)
            rot, trans = _PoseHeadFn.apply(
                x.reshape(-1, x.shape[-1]).float().contiguous(), self.fc_layers[0].weight,
                self.fc_layers[0].bias, self.fc_layers[2].weight, self.fc_layers[2].bias, self.rot_head.weight,
                self.rot_head.bias, self.trans_head.weight, self.trans_head.bias)
            return rot.view(*lead, 4), trans.view(*lead, 3)
        hidden = self.fc_layers(x)  # un-normalised quaternions (no shipped config): library ops
        rot = self.rot_head(hidden)
        if self.norm_rot:
            rot = F.normalize(rot, p=2, dim=-1)
        return rot, self.trans_head(hidden)


class StocasticPoseRegressor(PoseRegressor):
    """Appends `noise_dim` standard-normal channels to the input (MoN sampling); spelling as upstream."""

    def __init__(self, feat_dim, noise_dim, rot_type="quat", norm_rot=True):
        super().__init__(feat_dim + noise_dim, rot_type, norm_rot)
        self.noise_dim = noise_dim

    def forward(self, x):
        if self.noise_dim == 0:
            return super().forward(x)
        if x.is_cuda and torch.cuda.is_current_stream_capturing():
            # inside a HIP-graph capture a host-to-device copy would become a memcpy node replaying ONE draw (or freed
            # pinned memory) for ever: draw on the device generator, whose offset torch advances on every replay
            # (same rule as RGLNet._init_gru_hidden)
            noise = torch.randn(*x.shape[:-1], self.noise_dim, device=x.device, dtype=x.dtype)
        else:
            # CPU generator draws as upstream; pinned + asynchronous copy (a pageable copy drains the stream first)
            noise = torch.randn(*x.shape[:-1], self.noise_dim, pin_memory=x.is_cuda).to(x.device, non_blocking=True).type_as(x)
        return super().forward(torch.cat([x, noise], dim=-1))
