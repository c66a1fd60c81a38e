"""Dev probe (GPU): does the fused loss depend on stale workspace contents?  Poison patterns x search modes."""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_part_assembly_amd import _lib, synthetic
from multi_part_assembly_amd.rotation import Rotation3D
dev = torch.device("cuda:0")
B, P, N = 32, 20, 1000
batch = synthetic.make_batch(B, P, N, seed=1234, device=dev)
pcs, v = batch["part_pcs"], batch["part_valids"]
qg, tg = Rotation3D(batch["part_quat"]).rot.contiguous(), batch["part_trans"].contiguous()
L = _lib.lib()
nf, ni = ctypes.c_int64(), ctypes.c_int64()
L.mpa_assembly_loss_workspace(B, P, N, ctypes.byref(nf), ctypes.byref(ni))
fws = torch.empty(nf.value, device=dev)
iws = torch.empty(ni.value, dtype=torch.int32, device=dev)
losses = torch.empty(5, B, device=dev)
ref = None
for trial in range(3):
    torch.manual_seed(trial)
    qp = torch.nn.functional.normalize(torch.randn(B, P, 4), dim=-1).to(dev)
    tp = (torch.randn(B, P, 3) * (0.05 + 0.3 * trial)).to(dev)
    for mode in ("brute", "grid"):
        os.environ["MPA_SHAPE_SEARCH"] = mode
        for name, fval, ival in (("zero", 0.0, 0), ("nan", float("nan"), -1), ("big", 3e38, 0x7fffffff), ("neg", -3e38, -2**31),
                                 ("small", 1e-30, 12345678)):
            fws.fill_(fval); iws.fill_(ival)
            st = L.mpa_assembly_loss_forward(_lib.ptr(pcs), _lib.ptr(v), _lib.ptr(qp), _lib.ptr(tp), _lib.ptr(qg), _lib.ptr(tg), B, P, N, 1, 0,
                                             _lib.ptr(fws), _lib.ptr(iws), _lib.ptr(losses), _lib.current_stream(dev))
            torch.cuda.synchronize()
            cur = losses.clone()
            if name == "zero" and mode == "brute":
                ref = cur
            same = torch.equal(cur, ref)
            print(trial, mode, name, "status", st, "identical to brute/zero:", same, "" if same else [float(x) for x in cur.sum(1)], flush=True)
