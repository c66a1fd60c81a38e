import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_part_assembly_amd import loss as L, synthetic
from multi_part_assembly_amd.rotation import Rotation3D
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
batch = synthetic.make_batch(B, 20, 1000, seed=1234, device=dev)
pcs, v = batch["part_pcs"], batch["part_valids"]
rg, tg = Rotation3D(batch["part_quat"]), batch["part_trans"]
qp = torch.nn.functional.normalize(torch.randn(B, 20, 4), dim=-1).to(dev).requires_grad_()
tp = (torch.randn(B, 20, 3) * 0.05).to(dev).requires_grad_()
def step():
    qp.grad = None; tp.grad = None
    terms = L.geometric_assembly_loss(pcs, tp, Rotation3D(qp), tg, rg, v, training=True)[0]
    tot = sum(x.sum() for x in terms.values())
    tot.backward()
    return tot.detach()
for _ in range(2): step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
torch.cuda.synchronize(); print("captured")
for i in range(3):
    g.replay(); torch.cuda.synchronize(); print("replay", i, float(out))
