"""Dev probe (GPU): phases of the fused-loss forward under the three search modes (MPA_SHAPE_SEARCH = brute | grid |
leaf), for the everyday / artifact part mixes and an untrained (random) vs a trained (2 % off the ground truth)
prediction.  Phases from the library's own events: order+pose, per-part Chamfer, whole-shape Chamfer, finalize."""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_part_assembly_amd import _lib, synthetic
from multi_part_assembly_amd.rotation import Rotation3D
from multi_part_assembly_amd.loss import part_order
dev = torch.device("cuda:0")
B, P, N = 32, 20, 1000
L = _lib.lib()
modes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["grid", "leaf"]
ONLY = os.environ.get("PROBE_ONLY", "")  # e.g. "everyday:untrained" (for a profiler run of one case)
for preset in ("everyday", "artifact"):
    if ONLY and not ONLY.startswith(preset):
        continue
    batch = synthetic.make_batch(B, P, N, seed=1234, preset=preset, device=dev)
    pcs, v = batch["part_pcs"], batch["part_valids"]
    qg, tg = Rotation3D(batch["part_quat"]).rot.contiguous(), batch["part_trans"].contiguous()
    nf, ni = ctypes.c_int64(), ctypes.c_int64()
    L.mpa_assembly_loss_workspace(B, P, N, ctypes.byref(nf), ctypes.byref(ni))
    fws = torch.empty(nf.value, device=dev); iws = torch.empty(ni.value, dtype=torch.int32, device=dev)
    losses = torch.empty(5, B, device=dev)
    torch.manual_seed(0)
    noise_q = torch.randn(B, P, 4, device=dev); noise_t = torch.randn(B, P, 3, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        order = part_order(pcs, v)
    e1.record(); torch.cuda.synchronize()
    print(f"{preset}: valid parts {int(v.sum())}, part_order {e0.elapsed_time(e1) / 5:.3f} ms", flush=True)
    for regime in ("untrained", "trained"):
        if ONLY and ONLY.split(":")[1] != regime:
            continue
        if regime == "untrained":
            qp = torch.nn.functional.normalize(noise_q, dim=-1).contiguous(); tp = (0.1 * noise_t).contiguous()
        else:
            qp = torch.nn.functional.normalize(qg + 0.02 * noise_q, dim=-1).contiguous(); tp = (tg + 0.01 * noise_t).contiguous()
        ref = None
        for mode in modes:
            os.environ["MPA_SHAPE_SEARCH"] = mode
            for with_order in ((False, True) if mode == "leaf" else (False,)):
                evs = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
                for e in evs: e.record()
                acc = [0.0] * 5
                reps = 6
                for it in range(reps + 1):
                    h = (ctypes.c_void_p * 7)(*[e.cuda_event for e in evs])
                    st = L.mpa_assembly_loss_forward_ordered(_lib.ptr(pcs), _lib.ptr(v), _lib.ptr(qp), _lib.ptr(tp), _lib.ptr(qg), _lib.ptr(tg),
                                                            B, P, N, 1, 0, _lib.ptr(order) if with_order else None, -1, _lib.ptr(fws), _lib.ptr(iws),
                                                            _lib.ptr(losses), h, _lib.current_stream(dev))
                    _lib.check(st, "fwd")
                    torch.cuda.synchronize()
                    if it > 0:
                        for k in range(4): acc[k] += evs[k].elapsed_time(evs[k + 1])
                        acc[4] += evs[0].elapsed_time(evs[4])
                pn = B * P * N
                idx = iws[: 4 * pn].clone()
                if ref is None: ref = (idx, losses.clone())
                same = bool(torch.equal(idx.view(4, B, P, N)[:, v.bool()], ref[0].view(4, B, P, N)[:, v.bool()]))
                dl = float((losses - ref[1]).abs().max())
                print(f"  {regime:9s} {mode:5s}{'+order' if with_order else '      '}: pose {acc[0]/reps:.3f}  part-cd {acc[1]/reps:.3f}  shape-cd {acc[2]/reps:.3f}  "
                      f"finalize {acc[3]/reps:.3f}  total {acc[4]/reps:.3f} ms   idx==first mode: {same}  max|dloss| {dl:.2e}", flush=True)
