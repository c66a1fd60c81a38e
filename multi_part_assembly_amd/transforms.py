"""Quaternion point transforms over the HIP pose kernels — mirror of the reference's
`qrot / qtransform / rot_pc / transform_pc` (multi_part_assembly/utils/transforms.py:75-109,199-244).

One fused kernel per call (csrc/pose.hip) instead of repeat_interleave + two Hamilton products; the
backward is a per-part reduction kernel.  Fully differentiable in q, t and the points.
"""
from __future__ import annotations

import torch

from . import _lib
from .rotation import Rotation3D

__all__ = ["qrot", "qtransform", "rot_pc", "transform_pc", "pose_apply"]


class _PoseApply(torch.autograd.Function):
    """out[m, n] = quaternion_apply(q[m], fill-or-pc[m, n]) (+ t[m])  for pc [M, N, 3]."""

    @staticmethod
    def forward(ctx, pc, quat, trans, mask, fill):
        M, N = pc.shape[0], pc.shape[1]
        out = torch.empty_like(pc)
        with torch.cuda.device(pc.device):
            st = _lib.lib().mpa_pose_apply_forward(
                _lib.ptr(pc), _lib.ptr(quat), _lib.ptr(trans), _lib.ptr(mask), float(fill), M, N,
                _lib.ptr(out), _lib.current_stream(pc.device))
        _lib.check(st, "mpa_pose_apply_forward")
        ctx.save_for_backward(pc, quat, mask)
        ctx.fill = float(fill)
        ctx.has_trans = trans is not None
        return out

    @staticmethod
    def backward(ctx, gout):
        pc, quat, mask = ctx.saved_tensors
        M, N = pc.shape[0], pc.shape[1]
        gout = gout.contiguous()
        need_pc, need_q, need_t = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        gq = torch.empty_like(quat)
        gt = torch.empty((M, 3), dtype=pc.dtype, device=pc.device) if (ctx.has_trans and need_t) else None
        gpc = torch.empty_like(pc) if need_pc else None
        with torch.cuda.device(pc.device):
            st = _lib.lib().mpa_pose_apply_backward(
                _lib.ptr(gout), _lib.ptr(pc), _lib.ptr(quat), _lib.ptr(mask), ctx.fill, M, N,
                _lib.ptr(gq), _lib.ptr(gt), _lib.ptr(gpc), _lib.current_stream(pc.device))
        _lib.check(st, "mpa_pose_apply_backward")
        return gpc, (gq if need_q else None), gt, None, None


def pose_apply(pc, quat, trans=None, mask=None, fill=0.0):
    """Apply per-part poses: pc [..., N, 3], quat [..., 4], trans [..., 3] | None, mask [...] | None.

    Parts with mask == 0 have their points replaced by `fill` before the transform (the padded-part
    fill of shape_cd_loss, reference utils/loss.py:173-175).  fp32, CUDA tensors only.
    """
    if not pc.is_cuda:
        raise RuntimeError("pose_apply: only CUDA (HIP) tensors are supported")
    lead = pc.shape[:-2]
    if quat.shape[:-1] != lead or quat.shape[-1] != 4 or pc.shape[-1] != 3:
        raise RuntimeError(f"pose_apply: shape mismatch pc {tuple(pc.shape)} quat {tuple(quat.shape)}")
    N = pc.shape[-2]
    f32 = torch.float32
    args = [pc.to(f32).reshape(-1, N, 3).contiguous(), quat.to(f32).reshape(-1, 4).contiguous(),
            None if trans is None else trans.to(f32).reshape(-1, 3).contiguous(),
            None if mask is None else mask.to(f32).reshape(-1).contiguous()]
    if args[0].shape[0] > 65535:
        raise RuntimeError("pose_apply: more than 65535 parts in one call")
    return _PoseApply.apply(*args, fill).reshape(pc.shape)


def _per_point_fallback(q, v):
    # q and v already have equal leading shape (one quaternion per point): treat every point as a
    # one-point part.  Rare (not used by the training step); still runs on the HIP kernel.
    out = pose_apply(v.reshape(-1, 1, 3), q.reshape(-1, 4))
    return out.reshape(v.shape)


def qrot(q, v):
    """Rotate v (*, 3) by quaternion(s) q; q may omit the points axis ([B,P,4] vs [B,P,N,3])."""
    if q.dim() == v.dim() - 1:
        return pose_apply(v, q)
    assert q.shape[:-1] == v.shape[:-1]
    return _per_point_fallback(q, v)


def qtransform(t, q, v):
    """Rotate by q then translate by t (reference transforms.py:90-109)."""
    assert t.shape[-1] == 3
    if q.dim() == v.dim() - 1 and t.dim() == v.dim() - 1:
        return pose_apply(v, q, t)
    if t.dim() == v.dim() - 1:
        t = t.unsqueeze(-2)
    return qrot(q, v) + t


def _unwrap(rot, rot_type):
    if rot_type is None:
        assert isinstance(rot, Rotation3D)
        return rot.rot, rot.rot_type
    assert isinstance(rot, torch.Tensor)
    return rot, rot_type


def rot_pc(rot, pc, rot_type=None):
    """Rotate a point cloud by a Rotation3D (or a raw tensor when `rot_type` is given)."""
    r, kind = _unwrap(rot, rot_type)
    if kind != "quat":
        raise NotImplementedError(f"{kind} is not supported")
    return qrot(r, pc)


def transform_pc(trans, rot, pc, rot_type=None):
    """Rotate then translate a point cloud (reference transforms.py:223-244)."""
    r, kind = _unwrap(rot, rot_type)
    if kind != "quat":
        raise NotImplementedError(f"{kind} is not supported")
    return qtransform(trans, r, pc)
