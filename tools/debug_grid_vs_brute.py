"""Dev probe (GPU): at bench size, do the grid-pruned and the brute-force whole-shape searches give the same
pose gradients (i.e. the same arg-mins) and the same per-sample losses?"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_part_assembly_amd import _lib, synthetic
from multi_part_assembly_amd.rotation import Rotation3D
dev = torch.device("cuda:0")
B, P, N = 32, 20, 1000
L = _lib.lib()
nf, ni = ctypes.c_int64(), ctypes.c_int64()
L.mpa_assembly_loss_workspace(B, P, N, ctypes.byref(nf), ctypes.byref(ni))
fws = torch.empty(nf.value, device=dev); iws = torch.empty(ni.value, dtype=torch.int32, device=dev)
go = torch.ones(5, B, device=dev)
for trial in range(6):
    batch = synthetic.make_batch(B, P, N, seed=1234 + trial, device=dev)
    pcs, v = batch["part_pcs"], batch["part_valids"]
    qg, tg = Rotation3D(batch["part_quat"]).rot.contiguous(), batch["part_trans"].contiguous()
    torch.manual_seed(trial)
    eps = [0.0, 0.02, 0.2, 1.0, 5.0, 100.0][trial]
    qp = torch.nn.functional.normalize(qg + eps * torch.randn(B, P, 4, device=dev), dim=-1).contiguous()
    tp = (tg + min(eps, 1.0) * 0.3 * torch.randn(B, P, 3, device=dev)).contiguous()
    res = {}
    for mode in ("brute", "grid"):
        os.environ["MPA_SHAPE_SEARCH"] = mode
        losses = torch.empty(5, B, device=dev); gq, gt = torch.empty_like(qp), torch.empty_like(tp)
        L.mpa_assembly_loss_forward(_lib.ptr(pcs), _lib.ptr(v), _lib.ptr(qp), _lib.ptr(tp), _lib.ptr(qg), _lib.ptr(tg), B, P, N, 1, 0,
                                    _lib.ptr(fws), _lib.ptr(iws), _lib.ptr(losses), _lib.current_stream(dev))
        L.mpa_assembly_loss_backward(_lib.ptr(go), _lib.ptr(pcs), _lib.ptr(v), _lib.ptr(qp), _lib.ptr(tp), _lib.ptr(qg), _lib.ptr(tg), B, P, N, 1,
                                     _lib.ptr(fws), _lib.ptr(iws), _lib.ptr(gq), _lib.ptr(gt), _lib.current_stream(dev))
        torch.cuda.synchronize()
        res[mode] = (losses.clone(), gq.clone(), gt.clone())
    a, b = res["brute"], res["grid"]
    print(f"trial {trial} eps {eps}: grads identical {torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])}; "
          f"shape-cd max rel diff {float(((a[0][2] - b[0][2]).abs() / a[0][2].abs().clamp_min(1e-12)).max()):.2e}; "
          f"other terms identical {torch.equal(a[0][[0, 1, 3, 4]], b[0][[0, 1, 3, 4]])}", flush=True)
