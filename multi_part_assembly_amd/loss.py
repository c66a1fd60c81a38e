"""Pose losses of the assembly training step — mirror of the reference's `utils/loss.py`
(multi_part_assembly/utils/loss.py:7-202), same names / arguments / [B] return shapes.

The heavy parts run on the HIP library: pose application (csrc/pose.hip) and the Chamfer searches
(csrc/chamfer.hip).  The O(B*P) reductions around them are a handful of torch ops on tiny tensors.
`rot1/rot2` are `Rotation3D` (quaternion) objects.
"""
from __future__ import annotations

import torch

from .chamfer import chamfer_distance
from .transforms import pose_apply, rot_pc

__all__ = ["trans_l2_loss", "rot_l2_loss", "rot_cosine_loss", "rot_points_l2_loss",
           "rot_points_cd_loss", "shape_cd_loss", "repulsion_cd_loss"]

PAD_FILL = 1e3  # coordinate given to the points of padded parts in shape_cd_loss (loss.py:173-175)


def _valid_mean(loss_per_part, valids):
    """Mean over the valid parts of each sample: [B, P] x [B, P] -> [B]."""
    v = valids.float().detach()
    return (loss_per_part * v).sum(1) / v.sum(1)


def _quat(rot):
    if rot.rot_type != "quat":
        raise NotImplementedError(f"loss not supported for {rot.rot_type}")
    return rot.rot


def trans_l2_loss(trans1, trans2, valids):
    """Squared L2 between translations, [B, P, 3] x2 -> [B]."""
    return _valid_mean((trans1 - trans2).square().sum(-1), valids)


def rot_l2_loss(rot1, rot2, valids):
    """min(|q1 - q2|^2, |q1 + q2|^2) since q and -q are the same rotation."""
    q1, q2 = _quat(rot1), _quat(rot2)
    per_part = torch.minimum((q1 - q2).square().sum(-1), (q1 + q2).square().sum(-1))
    return _valid_mean(per_part, valids)


def rot_cosine_loss(rot1, rot2, valids):
    """1 - |<q1, q2>| per part."""
    q1, q2 = _quat(rot1), _quat(rot2)
    return _valid_mean(1.0 - (q1 * q2).sum(-1).abs(), valids)


def rot_points_l2_loss(pts, rot1, rot2, valids, ret_pts=False):
    """Mean squared distance between the two rotated copies of each part's points."""
    pts1, pts2 = rot_pc(rot1, pts), rot_pc(rot2, pts)
    loss = _valid_mean((pts1 - pts2).square().sum(-1).mean(-1), valids)
    return (loss, pts1, pts2) if ret_pts else loss


def rot_points_cd_loss(pts, rot1, rot2, valids, ret_pts=False):
    """Per-part Chamfer distance between the two rotated copies: B*P searches of N x N."""
    B, P = pts.shape[:2]
    pts1, pts2 = rot_pc(rot1, pts), rot_pc(rot2, pts)
    dist1, dist2 = chamfer_distance(pts1.flatten(0, 1), pts2.flatten(0, 1))
    per_part = (dist1.mean(1) + dist2.mean(1)).view(B, P).type_as(pts)
    loss = _valid_mean(per_part, valids)
    return (loss, pts1, pts2) if ret_pts else loss


def shape_cd_loss(pts, trans1, trans2, rot1, rot2, valids, ret_pts=False, training=True):
    """Whole-shape Chamfer distance after applying both pose sets: B searches of (P*N) x (P*N).

    The points of padded parts are set to PAD_FILL before the transform so they never match a real
    point (done inside the pose kernel, no clone of `pts`); no gradient flows into `pts`.
    training=True divides by all P*N slots ("hard negative mining", reference loss.py:185-193),
    training=False averages per part and then over the valid parts (:194-198).
    """
    B, P, N, _ = pts.shape
    pts = pts.detach()
    pts1 = pose_apply(pts, _quat(rot1), trans1, mask=valids, fill=PAD_FILL)
    pts2 = pose_apply(pts, _quat(rot2), trans2, mask=valids, fill=PAD_FILL)
    dist1, dist2 = chamfer_distance(pts1.flatten(1, 2), pts2.flatten(1, 2))
    v = valids.float().detach()
    if training:
        slot = v[:, :, None].expand(B, P, N).reshape(B, P * N)
        loss = (dist1 * slot).mean(1) + (dist2 * slot).mean(1)
    else:
        loss = _valid_mean((dist1 + dist2).view(B, P, N).mean(-1), v)
    return (loss, pts1, pts2) if ret_pts else loss


def repulsion_cd_loss(part_pcs, valids, thre):
    """max(0, thre - CD(part_i, part_j)) averaged over valid pairs (unused by the shipped configs)."""
    B, P, N, _ = part_pcs.shape
    a = part_pcs[:, :, None].expand(B, P, P, N, 3).reshape(-1, N, 3)
    b = part_pcs[:, None].expand(B, P, P, N, 3).reshape(-1, N, 3)
    dist1, dist2 = chamfer_distance(a.contiguous(), b.contiguous())
    cd = (thre - (dist1.mean(1) + dist2.mean(1)).view(B, P, P)).clamp_min(0.0)
    pair = (valids[:, :, None] * valids[:, None, :]).type_as(cd)
    return (cd * pair).sum([1, 2]) / pair.sum([1, 2])
