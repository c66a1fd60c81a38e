import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_part_assembly_amd import _lib, synthetic
from multi_part_assembly_amd.rotation import Rotation3D
dev = torch.device("cuda:0")
B, P, N = 32, 20, 1000
batch = synthetic.make_batch(B, P, N, seed=1234, device=dev)
pcs, v = batch["part_pcs"], batch["part_valids"]
qg, tg = Rotation3D(batch["part_quat"]).rot.contiguous(), batch["part_trans"].contiguous()
torch.manual_seed(0)
qp = torch.nn.functional.normalize(torch.randn(B, P, 4), dim=-1).to(dev)
tp = (torch.randn(B, P, 3) * 0.05).to(dev)
L = _lib.lib()
nf, ni = ctypes.c_int64(), ctypes.c_int64()
L.mpa_assembly_loss_workspace(B, P, N, ctypes.byref(nf), ctypes.byref(ni))
G = 1 << 20
fbuf = torch.full((nf.value + 2 * G,), 12345.0, device=dev)
ibuf = torch.full((ni.value + 2 * G,), 0x5A5A5A5A, dtype=torch.int32, device=dev)
fws, iws = fbuf[G:G + nf.value], ibuf[G:G + ni.value]
losses = torch.empty(5, B, device=dev)
gq, gt = torch.empty_like(qp), torch.empty_like(tp)
go = torch.ones(5, B, device=dev)
for mode in ("brute", "grid"):
    os.environ["MPA_SHAPE_SEARCH"] = mode
    st = L.mpa_assembly_loss_forward(_lib.ptr(pcs), _lib.ptr(v), _lib.ptr(qp), _lib.ptr(tp), _lib.ptr(qg), _lib.ptr(tg), B, P, N, 1, 0, _lib.ptr(fws), _lib.ptr(iws), _lib.ptr(losses), _lib.current_stream(dev))
    st2 = L.mpa_assembly_loss_backward(_lib.ptr(go), _lib.ptr(pcs), _lib.ptr(v), _lib.ptr(qp), _lib.ptr(tp), _lib.ptr(qg), _lib.ptr(tg), B, P, N, 1, _lib.ptr(fws), _lib.ptr(iws), _lib.ptr(gq), _lib.ptr(gt), _lib.current_stream(dev))
    torch.cuda.synchronize()
    bad_f = int((fbuf[:G] != 12345.0).sum() + (fbuf[G + nf.value:] != 12345.0).sum())
    bad_i = int((ibuf[:G] != 0x5A5A5A5A).sum() + (ibuf[G + ni.value:] != 0x5A5A5A5A).sum())
    print(mode, "status", st, st2, "guard violations float", bad_f, "int", bad_i, "loss", float(losses.sum()), "gq", float(gq.abs().sum()))
