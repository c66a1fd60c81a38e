"""Pose losses of the assembly training step — mirror of the reference's `utils/loss.py`
(multi_part_assembly/utils/loss.py:7-202), same names / arguments / [B] return shapes.

The heavy parts run on the HIP library: pose application (csrc/pose.hip) and the Chamfer searches
(csrc/chamfer.hip).  The O(B*P) reductions around them are a handful of torch ops on tiny tensors.
`rot1/rot2` are `Rotation3D` (quaternion) objects.
"""
from __future__ import annotations

import torch

import ctypes

from . import _lib
from .chamfer import chamfer_distance
from .transforms import pose_apply, rot_pc

__all__ = ["geometric_assembly_loss", "part_order", "search_mode", "SEARCH_MODES", "LOSS_TERMS", "trans_l2_loss", "rot_l2_loss", "rot_cosine_loss", "rot_points_l2_loss",
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


# ---- fused path ------------------------------------------------------------------------------------
LOSS_TERMS = ("trans_loss", "rot_pt_cd_loss", "transform_pt_cd_loss", "rot_loss", "rot_pt_l2_loss")


class LossTerms(dict):
    """{term name: [B]} that also carries the [K, B] tensor the terms are rows of (`stacked = (names, tensor)`):
    `BaseModel.loss_function` weights that tensor directly — selecting five rows and stacking them again costs
    ~15 small launches per step in autograd (five zero-filled [K, B] gradients, copies and adds)."""

    stacked = None


def part_order(part_pcs, valids):
    """The k-d order of a batch's parts that both Chamfer searches of the fused loss run on (csrc/leaf_nn.hip,
    `mpa_assembly_order`): a function of `part_pcs` and `valids` only — not of any pose — so ONE ordering serves every
    loss evaluation of a step (the GNN iterations of DGL / RGL-NET, the min-of-N samples, both directions).  Returns a
    float32 tensor to pass as `geometric_assembly_loss(..., order=...)`, or None where the loss keeps its grid search
    (N > 2048).  Only steers speed: the loss is bit-identical with or without it."""
    if not part_pcs.is_cuda:
        raise RuntimeError("part_order: only CUDA (HIP) tensors are supported")
    B, P, N, _ = part_pcs.shape
    L = _lib.lib()
    ne = ctypes.c_int64()
    _lib.check(L.mpa_assembly_order_elems(B, P, N, ctypes.byref(ne)), "mpa_assembly_order_elems")
    if ne.value == 0:
        return None
    pcs = part_pcs.detach().to(torch.float32).contiguous()
    v = valids.detach().to(torch.float32).contiguous()
    order = torch.empty(ne.value, dtype=torch.float32, device=pcs.device)
    with torch.cuda.device(pcs.device):
        st = L.mpa_assembly_order(_lib.ptr(pcs), _lib.ptr(v), B, P, N, _lib.ptr(order), _lib.current_stream(pcs.device))
    _lib.check(st, "mpa_assembly_order")
    return order


class _AssemblyLoss(torch.autograd.Function):
    """All five geometric loss terms in 5 launches forward / 1 launch backward (csrc/assembly_loss.hip)."""

    @staticmethod
    def forward(ctx, part_pcs, valids, quat_pred, trans_pred, quat_gt, trans_gt, training, fill_pads, order=None,
                search=-1):
        B, P, N, _ = part_pcs.shape
        dev = part_pcs.device
        L = _lib.lib()
        nf, ni = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(L.mpa_assembly_loss_workspace(B, P, N, ctypes.byref(nf), ctypes.byref(ni)),
                   "mpa_assembly_loss_workspace")
        fws = torch.empty(nf.value, dtype=torch.float32, device=dev)
        iws = torch.empty(ni.value, dtype=torch.int32, device=dev)
        losses = torch.empty((5, B), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            # library-recorded events: [0..4] bound the four phases, [5] / [6] sit right around the grid search kernel
            tag = f"[{B}x{P}x{N}]"
            names = [k + tag for k in ("assembly_pose", "assembly_part_chamfer", "assembly_shape_chamfer",
                                       "assembly_finalize")]
            evs = _lib.KernelTimer.phase_events(names)
            gs = _lib.KernelTimer.phase_events(["shape_search_kernel" + tag])
            both = None
            if evs is not None or gs is not None:
                both = (evs if evs is not None else [None] * 5) + (gs if gs is not None else [None] * 2)
            st = L.mpa_assembly_loss_forward_ordered(
                _lib.ptr(part_pcs), _lib.ptr(valids), _lib.ptr(quat_pred), _lib.ptr(trans_pred),
                _lib.ptr(quat_gt), _lib.ptr(trans_gt), B, P, N, int(training), int(fill_pads),
                _lib.ptr(order) if order is not None else None, int(search),
                _lib.ptr(fws), _lib.ptr(iws), _lib.ptr(losses), _lib.KernelTimer.handles(both),
                _lib.current_stream(dev))
            _lib.KernelTimer.add_phases(names, evs)
            _lib.KernelTimer.add_phases(["shape_search_kernel" + tag], gs)
        _lib.check(st, "mpa_assembly_loss_forward")
        ctx.save_for_backward(part_pcs, valids, quat_pred, trans_pred, quat_gt, trans_gt, fws, iws)
        ctx.training = int(training)
        cloud = B * P * N * 3
        pts = fws[: 4 * cloud].view(4, B, P, N, 3)
        ctx.mark_non_differentiable(pts)
        ctx.set_materialize_grads(False)   # else every backward zero-fills a [4, B, P, N, 3] "gradient" for pts
        return losses, pts

    @staticmethod
    def backward(ctx, grad_losses, _grad_pts):
        if getattr(ctx, "consumed", False):
            raise RuntimeError("assembly loss: the backward pass reuses the forward's tile-sum area as scratch: a second "
                               "backward over the same forward (retain_graph=True) is not supported — run the forward again")
        if grad_losses is None:
            return (None,) * 10
        ctx.consumed = True
        part_pcs, valids, quat_pred, trans_pred, quat_gt, trans_gt, fws, iws = ctx.saved_tensors
        B, P, N, _ = part_pcs.shape
        dev = part_pcs.device
        gq = torch.empty_like(quat_pred)
        gt = torch.empty_like(trans_pred)
        grad_losses = grad_losses.contiguous()
        with torch.cuda.device(dev):
            st = _lib.lib().mpa_assembly_loss_backward(
                _lib.ptr(grad_losses), _lib.ptr(part_pcs), _lib.ptr(valids), _lib.ptr(quat_pred),
                _lib.ptr(trans_pred), _lib.ptr(quat_gt), _lib.ptr(trans_gt), B, P, N, ctx.training,
                _lib.ptr(fws), _lib.ptr(iws), _lib.ptr(gq), _lib.ptr(gt), _lib.current_stream(dev))
        _lib.check(st, "mpa_assembly_loss_backward")
        return None, None, gq, gt, None, None, None, None, None, None


SEARCH_MODES = {"brute": 0, "grid": 1, "leaf": 2, "auto": 3}


class _LossReduce(torch.autograd.Function):
    """loss = sum_k w_k mean_b terms[k, b] and the per-term means, one launch each way (csrc/pose.hip
    mpa_loss_reduce_*; the reference: base_model.py:348-387 with one stochastic sample)."""

    @staticmethod
    def forward(ctx, terms, weights):
        terms = terms.contiguous()
        K, B = terms.shape
        means = torch.empty(K, dtype=torch.float32, device=terms.device)
        loss = torch.empty((), dtype=torch.float32, device=terms.device)
        with torch.cuda.device(terms.device):
            st = _lib.lib().mpa_loss_reduce_forward(_lib.ptr(terms), _lib.ptr(weights), K, B, _lib.ptr(means),
                                                    _lib.ptr(loss), _lib.current_stream(terms.device))
        _lib.check(st, "mpa_loss_reduce_forward")
        ctx.save_for_backward(weights)
        ctx.shape = (K, B)
        ctx.set_materialize_grads(False)
        return means, loss

    @staticmethod
    def backward(ctx, g_means, g_loss):
        if g_means is None and g_loss is None:
            return None, None
        (weights,) = ctx.saved_tensors
        K, B = ctx.shape
        g_terms = torch.empty((K, B), dtype=torch.float32, device=weights.device)
        gm = g_means.contiguous() if g_means is not None else None
        gl = g_loss.contiguous() if g_loss is not None else None
        with torch.cuda.device(weights.device):
            st = _lib.lib().mpa_loss_reduce_backward(_lib.ptr(gl) if gl is not None else None,
                                                     _lib.ptr(gm) if gm is not None else None, _lib.ptr(weights), K, B,
                                                     _lib.ptr(g_terms), _lib.current_stream(weights.device))
        _lib.check(st, "mpa_loss_reduce_backward")
        return g_terms, None


def weighted_term_means(terms, weights):
    """terms [K, B], weights [K] (float32, same CUDA device) -> (means [K], loss scalar) through the library."""
    return _LossReduce.apply(terms, weights)


def search_mode(name=None):
    """The search behind the loss's two Chamfer terms as the C ABI's code: MPA_SHAPE_SEARCH if set (the override for tests
    and A/B runs), else `name` (a key of SEARCH_MODES; what a configuration asks for), else "grid".  Identical results all;
    "leaf" / "auto" pay where parts are many and small and need the batch's `part_order`."""
    import os
    name = os.environ.get("MPA_SHAPE_SEARCH") or name or "grid"
    if name not in SEARCH_MODES:
        raise ValueError(f"shape search must be one of {sorted(SEARCH_MODES)}, not {name!r}")
    return SEARCH_MODES[name]


def geometric_assembly_loss(part_pcs, pred_trans, pred_rot, gt_trans, gt_rot, valids, training=True,
                            ret_pts=False, order=None, search=None):
    """The loss terms of `BaseModel._calc_loss` for geometric data, fused (no GT re-matching).

    Returns ({name: [B]} for LOSS_TERMS, pts) where pts is None or, with ret_pts,
    (pred_trans_pts, gt_trans_pts) [B,P,N,3] as shape_cd_loss(..., ret_pts=True) returns them.
    Gradients flow to pred_trans and pred_rot only (the GT pose is detached, as in the reference).
    `search`: "brute" | "grid" | "leaf" | "auto" (None: MPA_SHAPE_SEARCH, else "grid"): which exact searches run the two
    Chamfer terms — identical results.  `order`: `part_order(part_pcs, valids)` of this batch for "leaf" / "auto" when the
    caller evaluates the loss more than once per batch (computed by the call itself when None).
    """
    if not part_pcs.is_cuda:
        raise RuntimeError("geometric_assembly_loss: only CUDA (HIP) tensors are supported")
    B, P, N, _ = part_pcs.shape
    f = lambda t: t.detach().to(torch.float32).contiguous()
    losses, pts = _AssemblyLoss.apply(
        f(part_pcs), f(valids), _quat(pred_rot).to(torch.float32).contiguous(),
        pred_trans.to(torch.float32).contiguous(), f(_quat(gt_rot)), f(gt_trans), bool(training),
        bool(ret_pts), order, search_mode(search))
    terms = LossTerms((name, losses[i]) for i, name in enumerate(LOSS_TERMS))
    terms.stacked = (LOSS_TERMS, losses)
    return terms, ((pts[2], pts[3]) if ret_pts else None)
