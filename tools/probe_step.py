"""Dev probe (GPU): isolate Chamfer timing on the real training clouds vs uniform random."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from multi_part_assembly_amd import chamfer as C, config, synthetic, transforms as TR
from multi_part_assembly_amd.rotation import Rotation3D
dev = torch.device("cuda:0")

def t(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

batch = synthetic.make_batch(32, 20, 1000, seed=1234, device=dev)
pcs, v = batch["part_pcs"], batch["part_valids"]
qg = Rotation3D(batch["part_quat"]).rot
qp = torch.nn.functional.normalize(torch.randn(32, 20, 4, device=dev), dim=-1)
tp = torch.randn(32, 20, 3, device=dev) * 0.3
p1 = TR.pose_apply(pcs, qp); p2 = TR.pose_apply(pcs, qg)
s1 = TR.pose_apply(pcs, qp, tp, mask=v, fill=1e3).flatten(1, 2).contiguous()
s2 = TR.pose_apply(pcs, qg, batch["part_trans"], mask=v, fill=1e3).flatten(1, 2).contiguous()
a, b = p1.flatten(0, 1).contiguous(), p2.flatten(0, 1).contiguous()
for var in (0, 1, 2):
    print(f"variant {var}: part-CD real {t(lambda: C.chamfer_forward(a, b, variant=var)):.3f} ms, "
          f"shape-CD real {t(lambda: C.chamfer_forward(s1, s2, variant=var)):.3f} ms")
ra, rb = torch.rand_like(a), torch.rand_like(b)
rs1, rs2 = torch.rand_like(s1), torch.rand_like(s2)
for var in (0, 1, 2):
    print(f"variant {var}: part-CD rand {t(lambda: C.chamfer_forward(ra, rb, variant=var)):.3f} ms, "
          f"shape-CD rand {t(lambda: C.chamfer_forward(rs1, rs2, variant=var)):.3f} ms")
# valid-only shape clouds (no padded duplicates), same sizes: replace padded points by random far points
far = torch.rand_like(s1) * 50 + 100
m = v[:, :, None].expand(32, 20, 1000).reshape(32, 20000, 1) > 0
n1, n2 = torch.where(m, s1, far), torch.where(m, s2, far)
print(f"variant 1: shape-CD real, pads replaced by scattered far points {t(lambda: C.chamfer_forward(n1, n2, variant=1)):.3f} ms")
