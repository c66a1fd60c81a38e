"""Device-side GT <-> prediction matching of equivalent parts (csrc/match.hip).

`linear_sum_assignment` mirrors scipy's (square problems, batched); `match_parts` does the work of
`BaseModel._match_parts` (multi_part_assembly/models/modules/base_model.py:181-238) for every group of every sample of
the batch in three launches, with no device-to-host copy."""
from __future__ import annotations

import torch

from . import _lib

SUBSAMPLE = 100  # points per part in the cost matrix (base_model.py:163)


def linear_sum_assignment(cost: torch.Tensor, sizes: torch.Tensor | None = None) -> torch.Tensor:
    """cost [problems, ld, ld] float32 on the HIP device (sizes[i] x sizes[i] used; default the full ld) ->
    col4row [problems, ld] int32, the column assigned to each row (scipy's `col_ind`; -1 past the size)."""
    if not cost.is_cuda:
        raise RuntimeError("linear_sum_assignment: only CUDA (HIP) tensors are supported")
    cost = cost.detach().to(torch.float32).contiguous()
    problems, ld, ld2 = cost.shape
    assert ld == ld2, "square problems only"
    dev = cost.device
    if sizes is None:
        sizes = torch.full((problems,), ld, dtype=torch.int32, device=dev)
    sizes = sizes.to(device=dev, dtype=torch.int32).contiguous()
    out = torch.empty((problems, ld), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        st = _lib.lib().mpa_linear_sum_assignment(_lib.ptr(cost), _lib.ptr(sizes), problems, ld, _lib.ptr(out),
                                                  _lib.current_stream(dev))
    _lib.check(st, "mpa_linear_sum_assignment")
    return out


def match_parts(part_pcs, pred_trans, pred_quat, gt_trans, gt_quat, match_ids, sample_idx, ret_aux=False):
    """GT poses rearranged inside every group of equivalent parts so that they line up with the predictions at
    minimum Chamfer cost.  match_ids [B,P] (0 = unique / padded, g >= 1 = group g), sample_idx [B,G,n] point
    indices per group slot.  Returns (new_trans [B,P,3], new_quat [B,P,4]) and, with ret_aux, also
    (perm [B,P], cost [B,G,P,P], col4row [B,G,P])."""
    if not part_pcs.is_cuda:
        raise RuntimeError("match_parts: only CUDA (HIP) tensors are supported")
    B, P, N, _ = part_pcs.shape
    dev = part_pcs.device
    f = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
    i32 = lambda t: t.detach().to(device=dev, dtype=torch.int32).contiguous()
    sample_idx = i32(sample_idx)
    _, G, n = sample_idx.shape
    cost = torch.empty((B, G, P, P), dtype=torch.float32, device=dev)
    col4row = torch.empty((B, G, P), dtype=torch.int32, device=dev)
    new_t = torch.empty((B, P, 3), dtype=torch.float32, device=dev)
    new_q = torch.empty((B, P, 4), dtype=torch.float32, device=dev)
    perm = torch.empty((B, P), dtype=torch.int32, device=dev)
    args = [f(part_pcs), f(pred_trans), f(pred_quat), f(gt_trans), f(gt_quat), i32(match_ids), sample_idx]
    with torch.cuda.device(dev):
        st = _lib.lib().mpa_match_parts(*[_lib.ptr(a) for a in args], B, P, N, G, n, _lib.ptr(cost),
                                        _lib.ptr(col4row), _lib.ptr(new_t), _lib.ptr(new_q), _lib.ptr(perm),
                                        _lib.current_stream(dev))
    _lib.check(st, "mpa_match_parts")
    return (new_t, new_q, perm, cost, col4row) if ret_aux else (new_t, new_q)
