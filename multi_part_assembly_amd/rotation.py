"""`Rotation3D` — quaternion-only mirror of the reference's rotation value type
(reference: multi_part_assembly/utils/rotation.py:91-309).

Scope: every shipped model config sets `rot_type='quat'`, so only the
quaternion representation is carried; asking for 'rmat'/'axis' raises NotImplementedError.
Semantics kept from the reference constructor (rotation.py:115-147): the tensor is cast to fp32
and quaternions whose norm is <= 0.5 (the all-zero rows of padded parts) are replaced by the
identity (1, 0, 0, 0); nothing is normalised.
"""
from __future__ import annotations

import torch

_TENSOR_METHODS = ("reshape", "view", "squeeze", "unsqueeze", "flatten", "unflatten", "transpose",
                   "permute", "contiguous", "to", "cuda", "type", "type_as", "detach", "clone")


_IDENT = {}


def _identity_quat(device):
    """(1, 0, 0, 0) on `device`, created once (the constructor runs several times per training step)."""
    key = (device.type, device.index)
    if key not in _IDENT:
        _IDENT[key] = torch.tensor([1.0, 0.0, 0.0, 0.0], device=device)
    return _IDENT[key]


class _SanitizeFn(torch.autograd.Function):
    """Quaternions with norm <= 0.5 -> identity (the constructor rule); the gradient passes where the input was kept."""

    @staticmethod
    def forward(ctx, rot):
        from . import _lib
        q = rot.contiguous()
        out = torch.empty_like(q)
        keep = torch.empty(q.shape[:-1] + (1,), dtype=torch.float32, device=q.device)
        with torch.cuda.device(q.device):
            st = _lib.lib().mpa_quat_sanitize(_lib.ptr(q), q.numel() // 4, _lib.ptr(out), _lib.ptr(keep),
                                              _lib.current_stream(q.device))
        _lib.check(st, "mpa_quat_sanitize")
        ctx.save_for_backward(keep)
        return out

    @staticmethod
    def backward(ctx, grad):
        (keep,) = ctx.saved_tensors
        return grad * keep


class Rotation3D:
    ROT_TYPE = ["quat"]

    def __init__(self, rot, rot_type="quat", _sanitized=False):
        """`_sanitized` (internal): `rot` is the float32 tensor of another Rotation3D passed through an operation that keeps
        every quaternion as it is (detach, clone, a change of device) — the constructor rule is idempotent, so applying it
        again would be one more launch for the same values."""
        if _sanitized and rot.dtype == torch.float32 and rot.shape[-1] == 4 and rot_type == "quat":
            self._rot, self._rot_type = rot, rot_type
            return
        if rot_type != "quat":
            raise NotImplementedError(
                f"rotation {rot_type!r}: only 'quat' is on the MI355X hot path (every shipped "
                "model config uses it)")
        assert isinstance(rot, torch.Tensor), "rotation must be a tensor"
        assert rot.shape[-1] == 4, "wrong quaternion shape"
        rot = rot.float()
        if rot.is_cuda:  # one HIP launch (csrc/pose.hip) instead of norm + compare + where
            self._rot = _SanitizeFn.apply(rot)
        else:
            with torch.no_grad():
                keep = rot.norm(p=2, dim=-1, keepdim=True) > 0.5
            self._rot = torch.where(keep, rot, _identity_quat(rot.device))  # [4] broadcasts over the batch
        self._rot_type = rot_type

    # --- value access -----------------------------------------------------------------------
    @property
    def rot(self):
        return self._rot

    @rot.setter
    def rot(self, value):
        self.__init__(value, self._rot_type)

    @property
    def rot_type(self):
        return self._rot_type

    def convert(self, rot_type):
        if rot_type != "quat":
            raise NotImplementedError(f"conversion to {rot_type!r} is outside the quaternion hot path")
        return self.clone()

    def to_quat(self):
        return self.convert("quat").rot

    def to_euler(self, order="zyx", to_degree=True):
        """Euler angles [..., 3] (reference rotation.py:201-204; only its default convention is provided)."""
        if order != "zyx" or not to_degree:
            raise NotImplementedError("only the 'zyx' / degree convention of the evaluation metrics is provided")
        from .eval_utils import quat_to_euler_zyx_deg
        return quat_to_euler_zyx_deg(self.to_quat())

    # --- tensor-like surface ------------------------------------------------------------------
    shape = property(lambda self: self._rot.shape)
    device = property(lambda self: self._rot.device)
    dtype = property(lambda self: self._rot.dtype)

    def __len__(self):
        return self._rot.shape[0]

    def __getitem__(self, key):
        return Rotation3D(self._rot[key], self._rot_type)

    @staticmethod
    def _combine(op, rot_lst, dim):
        assert isinstance(rot_lst, (list, tuple)) and all(isinstance(r, Rotation3D) for r in rot_lst)
        return Rotation3D(op([r.rot for r in rot_lst], dim=dim), rot_lst[0].rot_type)

    @staticmethod
    def cat(rot_lst, dim=0):
        return Rotation3D._combine(torch.cat, rot_lst, dim)

    @staticmethod
    def stack(rot_lst, dim=0):
        return Rotation3D._combine(torch.stack, rot_lst, dim)


_VALUE_PRESERVING = ("detach", "clone", "contiguous", "cuda", "cpu")  # every quaternion of the result is one of the input


def _delegate(name):
    def method(self, *args, **kwargs):
        return Rotation3D(getattr(self._rot, name)(*args, **kwargs), self._rot_type, _sanitized=name in _VALUE_PRESERVING)

    method.__name__ = name
    method.__doc__ = f"torch.Tensor.{name} applied to the wrapped quaternion tensor."
    return method


for _name in _TENSOR_METHODS:
    setattr(Rotation3D, _name, _delegate(_name))
