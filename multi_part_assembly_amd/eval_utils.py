"""Evaluation metrics of the assembly task — mirrors of the reference's utils/eval_utils.py:12-199 (`calc_part_acc`,
`calc_connectivity_acc`, `trans_metrics`, `rot_metrics`), same names, arguments and [B]
outputs.  Part accuracy runs the per-part Chamfer search on the HIP operator; the rest are small masked reductions.

`calc_connectivity_acc` gathers the contacting pairs with one `nonzero` instead of the reference's B*P*P Python loop
(eval_utils.py:84-96); the pair order — (b, i, j) ascending — is the same, and the result does not depend on it."""
from __future__ import annotations

import math

import torch

from .chamfer import chamfer_distance
from .transforms import transform_pc


def _valid_mean(per_part, valids):
    valids = valids.float().detach()
    return (per_part * valids).sum(1) / valids.sum(1)


@torch.no_grad()
def calc_part_acc(pts, trans1, trans2, rot1, rot2, valids):
    """Fraction of valid parts whose Chamfer distance between the two posed copies is below 0.01 -> [B]."""
    B, P = pts.shape[:2]
    pts1 = transform_pc(trans1, rot1, pts).flatten(0, 1)
    pts2 = transform_pc(trans2, rot2, pts).flatten(0, 1)
    dist1, dist2 = chamfer_distance(pts1, pts2)
    per_part = (dist1.mean(dim=1) + dist2.mean(dim=1)).view(B, P).type_as(pts)
    ok = (per_part < 0.01) & (valids == 1)
    return ok.sum(-1) / (valids == 1).sum(-1)


def _symmetric_copies(points):
    """The 8 sign flips of the xyz coordinates, in the reference's order (x outermost): [n, 3] -> [n, 8, 3]."""
    signs = torch.tensor([[sx, sy, sz] for sx in (1.0, -1.0) for sy in (1.0, -1.0) for sz in (1.0, -1.0)],
                         dtype=points.dtype, device=points.device)
    return points[:, None, :] * signs[None]


@torch.no_grad()
def calc_connectivity_acc(trans, rot, contact_points):
    """Fraction of annotated contacts (contact_points[b, i, j, 0] == 1) whose two contact points, moved by the
    predicted poses of parts i and j, come closer than 0.01 (squared distance, minimum over the 8x8 symmetric
    copies) -> the batch-wide value tiled to [B]."""
    B = trans.shape[0]
    rot_type, rot = rot.rot_type, rot.rot
    b, i, j = torch.nonzero(contact_points[..., 0] == 1, as_tuple=True)
    p1 = _symmetric_copies(contact_points[b, i, j, 1:])
    p2 = _symmetric_copies(contact_points[b, j, i, 1:])
    p1 = transform_pc(trans[b, i], rot[b, i], p1, rot_type=rot_type)
    p2 = transform_pc(trans[b, j], rot[b, j], p2, rot_type=rot_type)
    dist = ((p1[:, :, None] - p2[:, None, :]) ** 2).sum(-1).flatten(1).min(-1)[0]
    acc = (dist < 0.01).sum().float() / float(dist.numel())
    return torch.ones(B).type_as(trans) * acc


@torch.no_grad()
def trans_metrics(trans1, trans2, valids, metric):
    assert metric in ("mse", "rmse", "mae")
    diff = trans1 - trans2
    if metric == "mae":
        per_part = diff.abs().mean(dim=-1)
    else:
        per_part = diff.pow(2).mean(dim=-1)
        if metric == "rmse":
            per_part = per_part ** 0.5
    return _valid_mean(per_part, valids)


def quat_to_euler_zyx_deg(q):
    """Real-first unit quaternions [..., 4] -> (x, y, z) Euler angles in degrees, 'zyx' convention — the default of
    the reference's `Rotation3D.to_euler` (utils/rotation.py:35-90,201-204)."""
    w, x, y, z = q.unbind(-1)
    ex = torch.atan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y))
    ey = torch.asin(torch.clamp(2 * (w * y - x * z), -1.0, 1.0))
    ez = torch.atan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z))
    return torch.stack((ex, ey, ez), dim=-1) * 180.0 / math.pi


@torch.no_grad()
def rot_metrics(rot1, rot2, valids, metric):
    """Euler-angle (degree) error with the wrap at 180 handled -> [B]."""
    assert metric in ("mse", "rmse", "mae")
    d = (quat_to_euler_zyx_deg(rot1.to_quat()) - quat_to_euler_zyx_deg(rot2.to_quat())).abs()
    d = torch.minimum(d, 360.0 - d)
    if metric == "mae":
        per_part = d.abs().mean(dim=-1)
    else:
        per_part = d.pow(2).mean(dim=-1)
        if metric == "rmse":
            per_part = per_part ** 0.5
    return _valid_mean(per_part, valids)
