"""Dev probe (GPU): the matrix-core gated search (csrc/gate_nn.hip) against the exhaustive scan — the operator on the
per-part call shape and the fused loss's forward with MPA_PART_SEARCH = scan | gate, untrained and near-GT poses."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_part_assembly_amd import chamfer as C, loss as L, synthetic
from multi_part_assembly_amd.rotation import Rotation3D
dev = torch.device("cuda:0")


def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


g = torch.Generator().manual_seed(21)
for B, n in ((640, 1000), (353, 1000), (640, 512), (64, 4000), (2000, 256)):
    a = (torch.rand(B, n, 3, generator=g) - 0.5).to(dev)
    b = (torch.rand(B, n, 3, generator=g) - 0.5).to(dev)
    o2, o4 = C.chamfer_forward(a, b, variant=2), C.chamfer_forward(a, b, variant=4)
    ok = all(torch.equal(x, y) for x, y in zip(o2, o4))
    print(f"operator [{B},{n},3]^2: scan {t(lambda: C.chamfer_forward(a, b, variant=2)):.3f} ms, "
          f"gate {t(lambda: C.chamfer_forward(a, b, variant=4)):.3f} ms, bit-equal {ok}", flush=True)

batch = synthetic.make_batch(32, 20, 1000, seed=1234, device=dev)
pcs, v = batch["part_pcs"], batch["part_valids"]
rg, tg = Rotation3D(batch["part_quat"]), batch["part_trans"]
gg = torch.Generator().manual_seed(0)
qp = torch.nn.functional.normalize(torch.randn(32, 20, 4, generator=gg), dim=-1).to(dev)
tp = (torch.randn(32, 20, 3, generator=gg) * 0.05).to(dev)
qp2 = torch.nn.functional.normalize(batch["part_quat"] + 0.05 * torch.randn(32, 20, 4, device=dev), dim=-1)
qp2 = torch.where(v[..., None] > 0, qp2, qp)
tp2 = batch["part_trans"] + 0.02 * torch.randn(32, 20, 3, device=dev)
for name, q_, t_ in (("untrained", qp, tp), ("near-GT", qp2, tp2)):
    res = {}
    for mode in ("scan", "gate"):
        os.environ["MPA_PART_SEARCH"] = mode
        fn = lambda: L.geometric_assembly_loss(pcs, t_, Rotation3D(q_), tg, rg, v, training=True)[0]
        terms = fn()
        res[mode] = ({k: x.detach().clone() for k, x in terms.items()}, t(fn))
    same = all(torch.allclose(res["scan"][0][k], res["gate"][0][k], rtol=1e-6, atol=0) for k in res["scan"][0])
    print(f"fused loss forward, {name}: scan {res['scan'][1]:.3f} ms, gate {res['gate'][1]:.3f} ms, terms equal to 1e-6: {same}",
          flush=True)
