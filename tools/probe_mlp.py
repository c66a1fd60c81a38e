"""Dev probe (GPU): one 3-layer edge MLP (12 800 pair rows) forward + backward: csrc/mlp.hip vs library ops."""
import os, sys, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_part_assembly_amd.gnn import _PairMLP
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = _PairMLP(256, 128).to(dev).train()
ref = copy.deepcopy(m)
a = torch.randn(32, 20, 128, device=dev, requires_grad=True)
w = torch.randn(640, 20, 128, device=dev)
def run(mod, hip):
    mod.MIN_ROWS = 1 if hip else 10 ** 9
    for _ in range(3):
        mod.zero_grad(); (mod.forward_pairs(a, a) * w).sum().backward()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        mod.zero_grad(); (mod.forward_pairs(a, a) * w).sum().backward()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10
print("edge MLP fwd+bwd: hip %.3f ms, library %.3f ms" % (run(m, True), run(ref, False)))
