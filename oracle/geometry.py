"""CPU restatement (torch, fp32) of the quaternion transforms and the loss functions of the
geometric-assembly training step.  TEST INFRASTRUCTURE — see oracle/__init__.py.

Every function cites the reference lines it restates.  The quaternion algebra lives in a
third-party dependency that is absent from /root/reference: **pytorch3d.transforms** (un-vendored;
version unpinned — docs/install.md:17-18 `conda install pytorch3d -c pytorch3d`, setup.py lists it
without a version).  Restated here from its published definition: real-first quaternions,
Hamilton product, `quaternion_apply(q, p) = (q * (0, p) * conj(q))[1:]` with NO normalisation of q
(so a non-unit q scales the point by |q|^2).  Parity at that boundary is pinned only through
tests/golden/transforms.npz (reference utils/transforms.py running on the same restatement) and a
scipy.spatial.transform cross-check — "parity unpinned" in the strict sense, see DESIGN.md.

Chamfer distances come from oracle/chamfer.py (C).  Gradients are obtained by autograd over these
definitions, with the Chamfer backward following chamfer_kernel.cu:199-208.
"""
from __future__ import annotations

import numpy as np
import torch

from . import chamfer as _oc


# ---- pytorch3d.transforms (restated) -------------------------------------------------------------
def quat_raw_multiply(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Hamilton product, real part first; term order as published (left-to-right sums)."""
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack(
        (
            aw * bw - ax * bx - ay * by - az * bz,
            aw * bx + ax * bw + ay * bz - az * by,
            aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw,
        ),
        -1,
    )


def quat_apply(q: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
    """(q * (0,p) * conj(q))[1:], q [...,4] and p [...,3] with equal leading shape."""
    p4 = torch.cat((p.new_zeros(p.shape[:-1] + (1,)), p), -1)
    conj = q * q.new_tensor([1.0, -1.0, -1.0, -1.0])
    return quat_raw_multiply(quat_raw_multiply(q, p4), conj)[..., 1:]


# ---- utils/rotation.py ---------------------------------------------------------------------------
def checked_quat(q: torch.Tensor) -> torch.Tensor:
    """Rotation3D.__init__ for rot_type='quat' (utils/rotation.py:115-128,135-147): cast to fp32;
    quaternions with norm <= 0.5 (padded parts are all-zero) become (1,0,0,0); no normalisation.  (float64 inputs stay
    float64: the anchor evaluations of the tests run the whole oracle in double.)"""
    q = q if q.dtype == torch.float64 else q.float()
    with torch.no_grad():
        keep = torch.norm(q, p=2, dim=-1, keepdim=True).abs() > 0.5
        ident = torch.zeros_like(q)
        ident[..., 0] = 1.0
    return torch.where(keep.expand_as(q), q, ident)


# ---- utils/transforms.py -------------------------------------------------------------------------
def rot_pc(q: torch.Tensor, pc: torch.Tensor) -> torch.Tensor:
    """rot_pc/qrot (utils/transforms.py:75-87,199-220): q [B,P,4] broadcast over pc [B,P,N,3]."""
    if q.dim() == pc.dim() - 1:
        q = q.unsqueeze(-2).expand(pc.shape[:-1] + (4,))
    return quat_apply(q, pc)


def transform_pc(t: torch.Tensor, q: torch.Tensor, pc: torch.Tensor) -> torch.Tensor:
    """transform_pc/qtransform (utils/transforms.py:90-109,223-244): rotate, then add t."""
    if t.dim() == pc.dim() - 1:
        t = t.unsqueeze(-2)
    return rot_pc(q, pc) + t


# ---- Chamfer as an autograd function over the C oracle ---------------------------------------------
class _ChamferFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        dt = torch.float64 if xyz1.dtype == torch.float64 else torch.float32
        a = xyz1.detach().to(dt).contiguous().numpy()
        b = xyz2.detach().to(dt).contiguous().numpy()
        d1, i1, d2, i2 = _oc.chamfer_forward(a, b)
        ctx.save = (a, b, i1, i2)
        return torch.from_numpy(d1), torch.from_numpy(d2)

    @staticmethod
    def backward(ctx, g1, g2):
        a, b, i1, i2 = ctx.save
        ga, gb = _oc.chamfer_backward(g1.contiguous().numpy(), g2.contiguous().numpy(), a, b, i1, i2)
        return torch.from_numpy(ga), torch.from_numpy(gb)


def chamfer_distance(xyz1: torch.Tensor, xyz2: torch.Tensor):
    """chamfer_distance (utils/chamfer/chamfer.py:36-64) for [B,n,3] inputs -> (dist1, dist2)."""
    return _ChamferFn.apply(xyz1, xyz2)


# ---- utils/loss.py -------------------------------------------------------------------------------
def valid_mean(loss_per_part: torch.Tensor, valids: torch.Tensor) -> torch.Tensor:
    """_valid_mean (utils/loss.py:7-19)."""
    v = valids.to(loss_per_part.dtype).detach()
    return (loss_per_part * v).sum(1) / v.sum(1)


def trans_l2_loss(t1, t2, valids):
    """utils/loss.py:22-35."""
    return valid_mean((t1 - t2).pow(2).sum(-1), valids)


def rot_l2_loss(q1, q2, valids):
    """utils/loss.py:38-56 (unused by the shipped configs; kept for the fixture)."""
    return valid_mean(torch.minimum((q1 - q2).pow(2).sum(-1), (q1 + q2).pow(2).sum(-1)), valids)


def rot_cosine_loss(q1, q2, valids):
    """utils/loss.py:59-86, quaternion branch: 1 - |<q1, q2>|."""
    return valid_mean(1.0 - torch.abs(torch.sum(q1 * q2, dim=-1)), valids)


def rot_points_l2_loss(pts, q1, q2, valids):
    """utils/loss.py:89-110."""
    p1, p2 = rot_pc(q1, pts), rot_pc(q2, pts)
    return valid_mean((p1 - p2).pow(2).sum(-1).mean(-1), valids)


def rot_points_cd_loss(pts, q1, q2, valids, ret_pts=False):
    """utils/loss.py:113-138: per-part Chamfer between the two rotated copies of each part."""
    B = pts.shape[0]
    p1, p2 = rot_pc(q1, pts), rot_pc(q2, pts)
    d1, d2 = chamfer_distance(p1.flatten(0, 1), p2.flatten(0, 1))
    per_part = (d1.mean(1) + d2.mean(1)).view(B, -1)
    loss = valid_mean(per_part, valids)
    return (loss, p1, p2) if ret_pts else loss


def shape_cd_loss(pts, t1, t2, q1, q2, valids, ret_pts=False, training=True):
    """utils/loss.py:141-202: whole-shape Chamfer; padded parts' points := 1e3 BEFORE the transform
    (:173-175); training divides by all P*N slots (:185-193), eval by the real part count (:194-198)."""
    B, P, N, _ = pts.shape
    pts = pts.detach().clone()
    pts = pts.masked_fill(valids[..., None, None] == 0, 1e3)
    p1, p2 = transform_pc(t1, q1, pts), transform_pc(t2, q2, pts)
    d1, d2 = chamfer_distance(p1.flatten(1, 2), p2.flatten(1, 2))
    v = valids.to(d1.dtype).detach()
    if training:
        vv = v.unsqueeze(2).repeat(1, 1, N).view(B, -1)
        loss = (d1 * vv).mean(1) + (d2 * vv).mean(1)
    else:
        loss = valid_mean((d1 + d2).view(B, P, N).mean(-1), v)
    return (loss, p1, p2) if ret_pts else loss


# ---- models/modules/base_model.py ----------------------------------------------------------------
GEOMETRIC_LOSS_CFG = {  # configs/_base_/models/loss/geometric_loss.py:17-27
    "trans_loss_w": 1.0,
    "rot_pt_cd_loss_w": 10.0,
    "transform_pt_cd_loss_w": 10.0,
    "use_rot_loss": True,
    "rot_loss_w": 0.2,
    "use_rot_pt_l2_loss": True,
    "rot_pt_l2_loss_w": 1.0,
}


def calc_loss_geometric(pred_quat, pred_trans, part_pcs, gt_quat, gt_trans, valids, loss_cfg=None,
                        training=True):
    """BaseModel._calc_loss for the geometric (non-semantic) datasets (base_model.py:240-314, no
    GT re-matching: :255-257) followed by the weighting of loss_function (:366-372) with
    sample_iter = 1.  Quaternions must already be `checked_quat`-ed.  Returns dict name -> [B]."""
    cfg = dict(GEOMETRIC_LOSS_CFG if loss_cfg is None else loss_cfg)
    gt_quat, gt_trans = gt_quat.detach(), gt_trans.detach()
    out = {
        "trans_loss": trans_l2_loss(pred_trans, gt_trans, valids),
        "rot_pt_cd_loss": rot_points_cd_loss(part_pcs, pred_quat, gt_quat, valids),
        "transform_pt_cd_loss": shape_cd_loss(part_pcs, pred_trans, gt_trans, pred_quat, gt_quat,
                                              valids, training=training),
    }
    if cfg["use_rot_loss"]:
        out["rot_loss"] = rot_cosine_loss(pred_quat, gt_quat, valids)
    if cfg["use_rot_pt_l2_loss"]:
        out["rot_pt_l2_loss"] = rot_points_l2_loss(part_pcs, pred_quat, gt_quat, valids)
    total = 0.0
    for k, v in out.items():
        total = total + v * cfg[k + "_w"]
    out["loss"] = total
    return out


def to_numpy(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
