import sys, torch
sys.path.insert(0, ".")
from multi_part_assembly_amd import synthetic
from multi_part_assembly_amd.encoder import build_encoder
dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = build_encoder("pointnet", 256).to(dev).train()
bt = synthetic.make_batch(32, 20, 1000, preset="everyday", seed=1234, device=dev)
x, v = bt["part_pcs"].flatten(0, 1), bt["part_valids"].flatten()
w = torch.randn(640, 256, device=dev)
xv = x[v.bool()].reshape(-1, 3).double()
ptp = xv.T @ xv
print("expected xx xy xz yy yz zz | sums:", [float(ptp[a, b]) for a, b in ((0,0),(0,1),(0,2),(1,1),(1,2),(2,2))], xv.sum(0).tolist())
for r in range(6):
    enc.zero_grad(set_to_none=True)
    out = enc.forward_parts(x, v)
    (out * w).sum().backward()
    print(enc.conv1.weight.grad.flatten()[:12].tolist())
