#!/usr/bin/env python3
"""Counters of the grid-pruned search (build_variants/gridstats.so: grid_nn.hip with -DMPA_GRID_STATS) on variations of the
whole-shape Chamfer call."""
import ctypes
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from multi_part_assembly_amd import _lib  # noqa: E402

_lib.LIB_PATH = ROOT / "build_variants" / (sys.argv[1] if len(sys.argv) > 1 else "gridstats.so")
from multi_part_assembly_amd import chamfer, synthetic  # noqa: E402
from multi_part_assembly_amd.transforms import pose_apply  # noqa: E402

dev = torch.device("cuda", 0)
B, P, N = 32, 20, 1000
L = _lib.lib()
L.mpa_debug_grid_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]


def stats(a, b, label):
    out = (ctypes.c_ulonglong * 24)()
    L.mpa_debug_grid_stats(out, 1)
    chamfer.chamfer_forward(a, b, variant=3)
    torch.cuda.synchronize()
    L.mpa_debug_grid_stats(out, 1)
    items, lanes, batches, cands, longc, ringb, nobound = list(out)[:7]
    print(f"{label}: items {items}, lanes/item {lanes / max(items, 1):.1f}, scan_batch calls/item {batches / max(items, 1):.1f}, "
          f"candidates/item {cands / max(items, 1):.0f} (long ranges {longc / max(items, 1):.0f}), outer-ring batches/item "
          f"{ringb / max(items, 1):.1f}, items without a bound after the seed {nobound}")
    t = list(out)[8:16]
    waves = max(t[7], 1)
    names = ["prologue", "item header", "seed", "rings 0-1", "outer rings", "pads + store"]
    print("   shader-clock ticks per wave: " + ", ".join(f"{n} {t[i] / waves:.0f}" for i, n in enumerate(names))
          + f"; whole wave {t[6] / waves:.0f}; waves {t[7]}, items per wave {items / waves:.2f}")


def build(num_parts=None, fill=1e3, seed=1234):
    batch = synthetic.make_batch(B, P, N, preset="everyday", seed=seed, device=dev, num_parts=num_parts)
    v, pts = batch["part_valids"], batch["part_pcs"]
    g = torch.Generator(device="cpu").manual_seed(99)
    q_far = torch.nn.functional.normalize(torch.randn(B, P, 4, generator=g), dim=-1).to(dev)
    t_far = (torch.rand(B, P, 3, generator=g) * 0.8 - 0.4).to(dev)
    q_gt = torch.where(v[..., None] > 0, batch["part_quat"], q_far.new_tensor([1.0, 0.0, 0.0, 0.0]))
    sh = lambda q, t: pose_apply(pts, q, t, mask=v, fill=fill).reshape(B, P * N, 3).contiguous()
    return sh(q_far, t_far), sh(q_gt, batch["part_trans"])


z1, z2 = build(num_parts=[11] * B)
stats(z1, z2, "C far, 11 valid + 9 padded")
stats(z1[:, :11000].contiguous(), z2[:, :11000].contiguous(), "D C without padded")
w1 = z1.clone()
w1[:, 11000:] = 1e3
stats(w1, z2, "E padded q == padded t")
stats(z1[:, 11000:].contiguous(), z2[:, 11000:].contiguous(), "G only the padded points")
