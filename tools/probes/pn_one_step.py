"""One forward + backward of the PointNet encoder at the benchmark size (prints whatever the library prints)."""
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
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    enc.zero_grad(set_to_none=True)
    out = enc.forward_parts(x, v)
    (out * w).sum().backward()
    torch.cuda.synchronize()
