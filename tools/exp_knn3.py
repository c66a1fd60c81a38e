"""Dev probe (GPU): C = 3 kNN, gated search (csrc/dg_knn3_gate.h) vs the exhaustive knn3_kernel at the benchmark size."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_part_assembly_amd.encoder import knn_exact
dev = torch.device("cuda:0")


def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


g = torch.Generator().manual_seed(3)
for n, N in ((353, 1000), (514, 1000), (353, 512)):
    x = (torch.rand(n, N, 3, generator=g) - 0.5) * torch.rand(n, 1, 3, generator=g) * 0.6
    rows = torch.cat([x.reshape(n * N, 3), torch.zeros(n * N, 1)], dim=1).to(dev).contiguous()
    res = {}
    for mode in ("scan", "gate"):
        os.environ["MPA_KNN3"] = mode
        out = knn_exact(rows, n, N, 3)
        res[mode] = (out.clone(), t(lambda: knn_exact(rows, n, N, 3)))
    print(f"knn3 {n} x {N}: scan {res['scan'][1]:.3f} ms, gate {res['gate'][1]:.3f} ms, index-equal "
          f"{torch.equal(res['scan'][0], res['gate'][0])}", flush=True)
