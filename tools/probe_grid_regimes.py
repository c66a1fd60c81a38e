"""Dev probe (GPU): cost of the fused-loss forward (grid search inside) as a function of how well the predicted
poses match the ground truth — the pruning radius of the search is the current nearest-neighbour distance."""
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
fws = torch.empty(nf.value, device=dev); iws = torch.empty(ni.value, dtype=torch.int32, device=dev)
losses = torch.empty(5, B, device=dev)
torch.manual_seed(0)
noise_q = torch.randn(B, P, 4, device=dev); noise_t = torch.randn(B, P, 3, device=dev)
for mode in ("grid", "brute"):
    os.environ["MPA_SHAPE_SEARCH"] = mode
    for eps in (0.0, 0.01, 0.03, 0.1, 0.3, 1.0, 100.0):
        qp = torch.nn.functional.normalize(qg + eps * noise_q, dim=-1).contiguous()
        tp = (tg + 0.3 * min(eps, 1.0) * noise_t).contiguous() if eps < 100 else (0.02 * noise_t).contiguous()
        if eps >= 100:
            qp = torch.nn.functional.normalize(noise_q, dim=-1).contiguous()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(4):
            if it == 1: e0.record()
            L.mpa_assembly_loss_forward(_lib.ptr(pcs), _lib.ptr(v), _lib.ptr(qp), _lib.ptr(tp), _lib.ptr(qg), _lib.ptr(tg), B, P, N, 1, 0,
                                        _lib.ptr(fws), _lib.ptr(iws), _lib.ptr(losses), _lib.current_stream(dev))
        e1.record(); torch.cuda.synchronize()
        print(f"{mode:5s} pose noise {eps:6.2f}: loss forward {e0.elapsed_time(e1) / 3:.3f} ms   shape-cd {float(losses[2].mean()):.5f}", flush=True)
