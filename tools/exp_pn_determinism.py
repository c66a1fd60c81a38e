"""Run-to-run determinism of the PointNet encoder's backward at the benchmark size: the same forward + backward repeated,
every parameter gradient compared bit for bit with the first run's.   python tools/exp_pn_determinism.py [reps=60]"""
import sys
import torch
sys.path.insert(0, ".")
from multi_part_assembly_amd import synthetic
from multi_part_assembly_amd.encoder import build_encoder

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = build_encoder("pointnet", 256).to(dev).train()
bt = synthetic.make_batch(32, 20, 1000, preset="everyday", seed=1234, device=dev)
x, v = bt["part_pcs"].flatten(0, 1), bt["part_valids"].flatten()
w = torch.randn(640, 256, device=dev)
ref = None
bad = {}
for r in range(reps):
    enc.zero_grad(set_to_none=True)
    out = enc.forward_parts(x, v)
    (out * w).sum().backward()
    g = {k: p.grad.clone() for k, p in enc.named_parameters()}
    if ref is None:
        ref = g
        continue
    for k in g:
        if not torch.equal(g[k], ref[k]):
            n = int((g[k] != ref[k]).sum())
            bad.setdefault(k, []).append((r, n, float((g[k] - ref[k]).abs().max())))
print("diverging tensors:", {k: (len(v), v[:3]) for k, v in bad.items()} if bad else "none")
