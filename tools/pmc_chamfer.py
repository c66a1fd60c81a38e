"""Dev probe (GPU): run only the whole-shape Chamfer launch a few times (for rocprofv3 --pmc)."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from multi_part_assembly_amd import chamfer as C
dev = torch.device("cuda:0")
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 2
g = torch.Generator().manual_seed(0)
a = torch.rand(32, 20000, 3, generator=g).to(dev)
b = torch.rand(32, 20000, 3, generator=g).to(dev)
for _ in range(3):
    C.chamfer_forward(a, b, variant=variant)
torch.cuda.synchronize()
