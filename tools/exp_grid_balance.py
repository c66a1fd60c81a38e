#!/usr/bin/env python3
"""Is the grid-pruned search limited by load imbalance between (sample, direction) pairs?  Times the generic whole-shape
call on the benchmark batch and on batches made of ONE of its samples replicated 32 times (perfectly balanced)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from multi_part_assembly_amd import chamfer, synthetic  # noqa: E402
from multi_part_assembly_amd.transforms import pose_apply  # noqa: E402

dev = torch.device("cuda", 0)
B, P, N = 32, 20, 1000


def timed(a, b, n=10):
    for _ in range(3):
        chamfer.chamfer_forward(a, b, variant=3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        chamfer.chamfer_forward(a, b, variant=3)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


batch = synthetic.make_batch(B, P, N, preset="everyday", seed=1234, device=dev)
v, pts = batch["part_valids"], batch["part_pcs"]
g = torch.Generator(device="cpu").manual_seed(99)
q_far = torch.nn.functional.normalize(torch.randn(B, P, 4, generator=g), dim=-1).to(dev)
t_far = (torch.rand(B, P, 3, generator=g) * 0.8 - 0.4).to(dev)
q_gt = torch.where(v[..., None] > 0, batch["part_quat"], q_far.new_tensor([1.0, 0.0, 0.0, 0.0]))
sh = lambda q, t: pose_apply(pts, q, t, mask=v, fill=1e3).reshape(B, P * N, 3).contiguous()
x1, x2 = sh(q_far, t_far), sh(q_gt, batch["part_trans"])
npart = batch["num_parts"]
full = timed(x1, x2)
print(f"benchmark batch ({sum(npart)} valid parts): {full:.3f} ms")
tot = 0.0
for b in sorted(range(B), key=lambda i: npart[i])[::6]:
    r1, r2 = x1[b:b + 1].expand(B, -1, -1).contiguous(), x2[b:b + 1].expand(B, -1, -1).contiguous()
    t = timed(r1, r2)
    print(f"  sample {b:2d} ({npart[b]:2d} parts) x 32: {t:.3f} ms  -> {t / 32 * 1e3:.1f} us per sample")
