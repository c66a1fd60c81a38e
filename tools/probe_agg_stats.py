"""Phase times inside dg_agg_bwd_kernel (needs the instrumented build: tools/build_variant.sh agg_stats dgcnn_enc.hip
-DMPA_AGG_STATS, copied over libmpa_hip.so on the GPU box): per block, cycles of phase A (loads + fixed-point scatter), of the passes (per
wave) and in total; passes and loop iterations per wave."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_part_assembly_amd import _lib
from multi_part_assembly_amd.encoder import build_encoder
dev = torch.device("cuda:0")
n, N, F = int(os.environ.get("PARTS", "353")), 1000, 128
torch.manual_seed(0)
enc = build_encoder("dgcnn", F).to(dev).train()
x = torch.randn(n, N, 3, device=dev) * 0.2
w = torch.randn(n, F, device=dev)
def step():
    for p in enc.parameters(): p.grad = None
    (enc(x) * w).sum().backward()
for _ in range(2): step()
torch.cuda.synchronize()
if not hasattr(_lib.lib(), "mpa_debug_agg_stats"):  # a regular build: just run the steps (for rocprofv3 around this script)
    for _ in range(5): step()
    torch.cuda.synchronize()
    sys.exit(0)
fn = _lib.lib().mpa_debug_agg_stats
fn.restype = ctypes.c_int
buf = (ctypes.c_ulonglong * 8)()
fn(buf, 1)
step(); torch.cuda.synchronize()
fn(buf, 1)
blocks, t_load, t_pass, t_tot, passes, iters = buf[0], buf[1], buf[2], buf[3], buf[4], buf[5]
waves = int(os.environ.get('WAVES', '16'))
print(f"blocks {blocks}: cycles per block: phase A {t_load / blocks:.0f}, passes (mean over waves) {t_pass / blocks / waves:.0f}, "
      f"total {t_tot / blocks:.0f}; passes per wave {passes / blocks / waves:.2f}, loop iterations per pass {iters / max(1, passes):.2f}, "
      f"cycles per pass {t_pass / max(1, passes):.0f}; phase A: loads + maxima {buf[6] / blocks:.0f}, + scatter {buf[7] / blocks:.0f}")
