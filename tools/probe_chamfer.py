"""Dev probe (GPU): parity of both Chamfer kernel variants vs the oracle + A/B timing.
Usage on the GPU box: python tools/probe_chamfer.py [--big]"""
import sys, time, json
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from multi_part_assembly_amd import chamfer as C
from oracle import chamfer as OC

dev = torch.device("cuda:0")
print("device", torch.cuda.get_device_name(0), torch.version.hip)
rng = np.random.default_rng(1)


def check(B, n1, n2, dtype=np.float32, scale=1.0):
    a = (rng.random((B, n1, 3)) * scale).astype(dtype)
    b = (rng.random((B, n2, 3)) * scale).astype(dtype)
    ref = OC.chamfer_forward(a, b)
    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    ok = True
    for v in ([0, 1, 2] if dtype == np.float32 else [None]):
        out = C.chamfer_forward(ta, tb, variant=v)
        torch.cuda.synchronize()
        for r, o, nm in zip(ref, out, ["dist1", "idx1", "dist2", "idx2"]):
            same = np.array_equal(r, o.cpu().numpy())
            ok &= same
            if not same:
                diff = (r != o.cpu().numpy()).sum()
                print(f"  MISMATCH B={B} n1={n1} n2={n2} variant={v} {nm}: {diff} entries")
    print(f"parity B={B} n1={n1} n2={n2} {dtype.__name__}: {'OK' if ok else 'FAIL'}")
    return ok


def timeit(B, n1, n2, iters=10):
    a = torch.rand(B, n1, 3, device=dev)
    b = torch.rand(B, n2, 3, device=dev)
    res = {}
    for v in (0, 1, 2):
        for _ in range(2):
            C.chamfer_forward(a, b, variant=v)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            C.chamfer_forward(a, b, variant=v)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        pairs = 2.0 * B * n1 * n2
        res[v] = ms
        print(f"time B={B} n1={n1} n2={n2} variant={v}: {ms:.3f} ms  {pairs/ms/1e6:.1f} Gpairs/s "
              f"alg {24.0*B*(n1+n2)/ms/1e6:.2f} GB/s")
    return res


allok = True
for shp in [(2, 64, 64), (3, 100, 77), (1, 1000, 1000), (1, 1100, 900), (5, 1, 9), (2, 7, 0), (4, 300, 1031)]:
    allok &= check(*shp)
allok &= check(2, 513, 300, np.float64)
# near-tie stress: coordinates on a coarse lattice -> many exact ties
a = (rng.integers(0, 4, (3, 500, 3)) * 0.25).astype(np.float32)
b = (rng.integers(0, 4, (3, 700, 3)) * 0.25).astype(np.float32)
ref = OC.chamfer_forward(a, b)
for v in (0, 1, 2):
    out = C.chamfer_forward(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev), variant=v)
    same = all(np.array_equal(r, o.cpu().numpy()) for r, o in zip(ref, out))
    allok &= same
    print(f"lattice ties variant={v}: {'OK' if same else 'FAIL'}")
# backward
a = rng.random((2, 300, 3)).astype(np.float32); b = rng.random((2, 200, 3)).astype(np.float32)
d1, i1, d2, i2 = OC.chamfer_forward(a, b)
g1 = rng.standard_normal((2, 300)).astype(np.float32); g2 = rng.standard_normal((2, 200)).astype(np.float32)
r1, r2 = OC.chamfer_backward(g1, g2, a, b, i1, i2)
T = lambda x: torch.from_numpy(x).to(dev)
o1, o2 = C.chamfer_backward(T(g1), T(g2), T(a), T(b), T(i1), T(i2))
err = max(np.abs(o1.cpu().numpy() - r1).max(), np.abs(o2.cpu().numpy() - r2).max())
print("backward max abs err", err)
allok &= err < 1e-5
print("ALL", "OK" if allok else "FAIL")
timeit(640, 1000, 1000)
timeit(32, 20000, 20000, iters=3)
timeit(400, 100, 100)
