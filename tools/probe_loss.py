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

for remap in ("1", "0"):
    os.environ["MPA_XCD_REMAP"] = remap
    for q in ("2", "4"):
        os.environ["MPA_ASSEMBLY_Q"] = q
        print(f"remap={remap} Q={q}: forward {t(fwd):.3f} ms, forward+backward {t(fwdbwd):.3f} ms")
