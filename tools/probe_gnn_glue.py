"""Launch times of the graph-network glue kernels (csrc/gnn_glue.hip) at the benchmark shapes, forward and backward."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
from multi_part_assembly_amd.gnn_ops import narrow_linear_relu, relation_head, relation_mean, pair_rows


def timed(name, fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    print(f"{name:50s} {a.elapsed_time(b) / reps * 1e3:8.1f} us")


def fwd_bwd(name, make):
    g = torch.randn_like(make())
    timed(name + " forward", make)
    timed(name + " forward + backward", lambda: make().backward(g))


dev = "cuda"
for R, K, N in ((640, 7, 256), (64, 7, 256), (640, 7, 64), (5120, 7, 256)):
    x = torch.randn(R, K, device=dev, requires_grad=True)
    w = torch.randn(N, K, device=dev, requires_grad=True)
    b = torch.randn(N, device=dev, requires_grad=True)
    fwd_bwd(f"narrow_linear_relu R={R} K={K} N={N}", lambda: narrow_linear_relu(x, w, b))
h = torch.randn(12800, 512, device=dev, requires_grad=True)
w = torch.randn(1, 512, device=dev, requires_grad=True)
b = torch.randn(1, device=dev, requires_grad=True)
m = (torch.rand(12800, device=dev) < 0.5).float()
fwd_bwd("relation_head 12800 x 512", lambda: relation_head(h, w, b, m))
e = torch.randn(32, 20, 20, 128, device=dev, requires_grad=True)
r = torch.rand(32, 20, 20, device=dev, requires_grad=True)
fwd_bwd("relation_mean 32 x 20 x 20 x 128", lambda: relation_mean(e, r))
a = torch.randn(32, 20, 128, device=dev, requires_grad=True)
fwd_bwd("pair_rows 32 x 20 x 128", lambda: pair_rows(a, a))
