"""Dev probe (GPU): DGCNN encoder forward + backward at the benchmark's part count (time and peak memory)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_part_assembly_amd import _lib
from multi_part_assembly_amd.encoder import build_encoder
dev = torch.device("cuda:0")
n, N, F = int(os.environ.get("PARTS", "352")), 1000, 128
torch.manual_seed(0)
enc = build_encoder("dgcnn", F).to(dev).train()
x = (torch.randn(n, N, 3, device=dev) * 0.2)
w = torch.randn(n, F, device=dev)
def step():
    for p in enc.parameters(): p.grad = None
    out = enc(x); (out * w).sum().backward()
for _ in range(2): step()
torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
timer = _lib.KernelTimer(); _lib.KernelTimer.active = timer
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): step()
e1.record(); torch.cuda.synchronize(); _lib.KernelTimer.active = None
print(f"DGCNN fwd+bwd n={n} N={N} F={F}: {e0.elapsed_time(e1) / 5:.2f} ms/iter, peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")
for k, v in timer.summary().items(): print("   ", k, round(v["avg_ms"], 3), "ms x", v["launches"] // 5)
