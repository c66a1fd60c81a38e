"""Dev probe (GPU): time the fused assembly loss (forward, forward+backward) on the bench batch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_part_assembly_amd import loss as L, synthetic
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
print("valid parts", sum(batch["num_parts"]), "sum n^2", sum(k * k for k in batch["num_parts"]))
pcs, v = batch["part_pcs"], batch["part_valids"]
rg, tg = Rotation3D(batch["part_quat"]), batch["part_trans"]
g = torch.Generator().manual_seed(0)
qp = torch.nn.functional.normalize(torch.randn(32, 20, 4, generator=g), dim=-1).to(dev).requires_grad_()
tp = (torch.randn(32, 20, 3, generator=g) * 0.05).to(dev).requires_grad_()

def fwd():
    return L.geometric_assembly_loss(pcs, tp, Rotation3D(qp), tg, rg, v, training=True)[0]

def fwdbwd():
    terms = fwd()
    sum(x.sum() for x in terms.values()).backward()

for mode in ("brute", "grid"):
    os.environ["MPA_SHAPE_SEARCH"] = mode
    print(f"shape search={mode}: forward {t(fwd):.3f} ms, forward+backward {t(fwdbwd):.3f} ms")

# "trained" regime: predicted poses close to the ground truth (both shapes overlap)
qp2 = torch.nn.functional.normalize(batch["part_quat"] + 0.05 * torch.randn(32, 20, 4, device=dev), dim=-1)
qp2 = torch.where(v[..., None] > 0, qp2, qp.detach()).requires_grad_()
tp2 = (batch["part_trans"] + 0.02 * torch.randn(32, 20, 3, device=dev)).requires_grad_()
def fwd2():
    return L.geometric_assembly_loss(pcs, tp2, Rotation3D(qp2), tg, rg, v, training=True)[0]
for mode in ("brute", "grid"):
    os.environ["MPA_SHAPE_SEARCH"] = mode
    print(f"near-GT poses, shape search={mode}: forward {t(fwd2):.3f} ms")
