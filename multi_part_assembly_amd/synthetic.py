"""On-device synthetic batches with the reference's data_dict contract
(multi_part_assembly/datasets/geometry_data.py:133-207), for benchmarks and tests — the Breaking-Bad
meshes are not available, and the metric is defined on synthetic B x P x 1000 x 3 part clouds.

Per sample: num_parts ~ U{min..max}; every valid part is N points uniform in an axis-aligned box
(half-extents ~ U(lo, hi)^3) around its centroid c ~ U(-0.4, 0.4)^3.  As in `__getitem__`
(:133-146) the part is re-centred (part_trans = c), rotated by a Haar-random rotation R
(part_pcs = R (x - c)) and part_quat is the inverse rotation, real part first, so that
transform_pc(part_trans, part_quat, part_pcs) re-assembles the shape.  Padded slots are all-zero
and valid parts come first (:121-126,175-177).
"""
from __future__ import annotations

import torch

from .transforms import pose_apply

PRESETS = {
    "everyday": dict(min_parts=2, max_parts=20, half_extent=(0.02, 0.3)),   # breaking_bad/everyday.py:13-14
    "artifact": dict(min_parts=12, max_parts=20, half_extent=(0.01, 0.1)),  # many small parts
}


def make_batch(batch_size, max_parts=20, num_points=1000, preset="everyday", seed=1234,
               device="cuda", num_parts=None):
    """Returns a data_dict of float32 CUDA tensors (+ `num_parts` as a host list).

    `seed` should differ per rank (1234 + rank)."""
    cfg = PRESETS[preset]
    dev = torch.device(device)
    g = torch.Generator(device="cpu").manual_seed(seed)
    B, P, N = batch_size, max_parts, num_points
    if num_parts is None:
        num_parts = torch.randint(cfg["min_parts"], min(cfg["max_parts"], P) + 1, (B,), generator=g).tolist()
    valids = torch.zeros(B, P)
    for b, k in enumerate(num_parts):
        valids[b, :k] = 1.0
    lo, hi = cfg["half_extent"]
    half = torch.rand(B, P, 1, 3, generator=g) * (hi - lo) + lo
    local = (torch.rand(B, P, N, 3, generator=g) * 2 - 1) * half
    local = local - local.mean(dim=2, keepdim=True)
    centroid = torch.rand(B, P, 3, generator=g) * 0.8 - 0.4
    q_rot = torch.nn.functional.normalize(torch.randn(B, P, 4, generator=g), dim=-1)  # Haar on S^3
    v = valids.to(dev)
    local = local.to(dev) * v[..., None, None]
    q_rot = q_rot.to(dev)
    part_pcs = pose_apply(local, q_rot) * v[..., None, None]
    part_quat = q_rot * q_rot.new_tensor([1.0, -1.0, -1.0, -1.0]) * v[..., None]
    ids = torch.arange(P, device=dev, dtype=torch.float32)[None].expand(B, P) * v
    return {
        "part_pcs": part_pcs.contiguous(),
        "part_trans": (centroid.to(dev) * v[..., None]).contiguous(),
        "part_quat": part_quat.contiguous(),
        "part_valids": v,
        "instance_label": torch.zeros(B, P, 0, device=dev),
        "part_label": torch.zeros(B, P, 0, device=dev),
        "part_ids": ids,
        "valid_matrix": v[:, :, None] * v[:, None, :],
        "num_parts": num_parts,
    }


def make_semantic_batch(batch_size, max_parts=2, num_points=1000, seed=1234, device="cuda", num_part_category=57):
    """Semantic-dataset (PartNet-like) stand-in for the plumbing configuration (P = 2, B = 4,
    `match_ids = [1, 1]`): every shape has `max_parts` geometrically identical parts (one cloud, different poses), so
    the GT <-> prediction matching has a real group to permute; `instance_label` is the one-hot part slot
    (partnet_data.py:163-208), `part_label` zero-width (not in the shipped `data_keys`)."""
    batch = make_batch(batch_size, max_parts, num_points, preset="everyday", seed=seed, device=device,
                       num_parts=[max_parts] * batch_size)
    dev = batch["part_pcs"].device
    B, P = batch_size, max_parts
    batch["part_pcs"] = batch["part_pcs"][:, :1].expand(B, P, num_points, 3).contiguous()
    batch["match_ids"] = torch.ones(B, P, dtype=torch.int64, device=dev)
    batch["instance_label"] = torch.eye(P, device=dev)[None].repeat(B, 1, 1)
    return batch
