"""Work statistics of the leaf search (needs the instrumented build: tools/build_variant.sh leaf_stats leaf_nn.hip
-DMPA_LEAF_STATS, copied over libmpa_hip.so on the GPU box): waves, leaf tests / scans per wave, the heaviest wave."""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_part_assembly_amd import _lib, synthetic
from multi_part_assembly_amd.rotation import Rotation3D
dev = torch.device("cuda:0")
B, P, N = 32, 20, 1000
L = _lib.lib()
fn = L.mpa_debug_leaf_stats
fn.restype = ctypes.c_int
buf = (ctypes.c_ulonglong * 64)()
os.environ["MPA_SHAPE_SEARCH"] = "leaf"
for preset in ("everyday", "artifact"):
    batch = synthetic.make_batch(B, P, N, seed=1234, preset=preset, device=dev)
    pcs, v = batch["part_pcs"], batch["part_valids"]
    qg, tg = Rotation3D(batch["part_quat"]).rot.contiguous(), batch["part_trans"].contiguous()
    nf, ni = ctypes.c_int64(), ctypes.c_int64()
    L.mpa_assembly_loss_workspace(B, P, N, ctypes.byref(nf), ctypes.byref(ni))
    fws = torch.empty(nf.value, device=dev); iws = torch.empty(ni.value, dtype=torch.int32, device=dev)
    losses = torch.empty(5, B, device=dev)
    torch.manual_seed(0)
    noise_q = torch.randn(B, P, 4, device=dev); noise_t = torch.randn(B, P, 3, device=dev)
    for regime in ("untrained", "trained"):
        if regime == "untrained":
            qp = torch.nn.functional.normalize(noise_q, dim=-1).contiguous(); tp = (0.1 * noise_t).contiguous()
        else:
            qp = torch.nn.functional.normalize(qg + 0.02 * noise_q, dim=-1).contiguous(); tp = (tg + 0.01 * noise_t).contiguous()
        fn(buf, 1)
        st = L.mpa_assembly_loss_forward(_lib.ptr(pcs), _lib.ptr(v), _lib.ptr(qp), _lib.ptr(tp), _lib.ptr(qg), _lib.ptr(tg), B, P, N, 1, 0,
                                         _lib.ptr(fws), _lib.ptr(iws), _lib.ptr(losses), _lib.current_stream(dev))
        torch.cuda.synchronize()
        fn(buf, 1)
        for sh, name in ((0, "part "), (1, "shape")):
            g = buf[32 * sh: 32 * sh + 32]
            w = max(1, g[0])
            hist = " ".join(f"{1 << k}:{g[8 + k]}" for k in range(12) if g[8 + k])
            print(f"{preset} {regime} {name}: wave searches {g[0]} tests/wave {g[1]/w:.1f} scans/wave {g[2]/w:.1f} parts/wave {g[3]/w:.1f} "
                  f"max scans {g[4]}  exact leaf scans {g[5]}  deferred waves {g[7]}  scans hist {hist}", flush=True)
