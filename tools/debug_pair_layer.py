import sys, copy, torch
sys.path.insert(0, "/root/repo")
from multi_part_assembly_amd.gnn import _PairMLP
dev = torch.device("cuda", 0)
for (B, P, F, same) in [(3, 5, 64, False), (32, 20, 128, True), (2, 33, 128, False)]:
    torch.manual_seed(B * 100 + P)
    mine = _PairMLP(2 * F, 128).to(dev).train()
    with torch.no_grad():
        for bn in (mine.bn1, mine.bn2, mine.bn3):
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_(0, 0.1)
    rows = copy.deepcopy(mine); rows.PAIR_LAYER = False
    ref64 = copy.deepcopy(mine).double(); ref64.MIN_ROWS = 10 ** 12
    a0 = torch.randn(B, P, F, device=dev); b0 = a0 if same else torch.randn(B, P, F, device=dev)
    w = torch.randn(B * P, P, 128, device=dev)
    res = []
    for mod, cast in ((mine, torch.float32), (rows, torch.float32), (ref64, torch.float64)):
        a = a0.to(cast).clone().requires_grad_(); b = a if same else b0.to(cast).clone().requires_grad_()
        out = mod.forward_pairs(a, b); (out * w.to(cast)).sum().backward()
        res.append((out.detach(), a.grad, None if same else b.grad, {k: p.grad for k, p in mod.named_parameters()}))
    rel = lambda x, y: float((x.double() - y.double()).abs().max() / (y.double().abs().max() + 1e-12))
    print((B, P, F, same), "out", rel(res[0][0], res[2][0]), "ga", rel(res[0][1], res[2][1]), rel(res[1][1], res[2][1]),
          "gb", None if same else (rel(res[0][2], res[2][2]), rel(res[1][2], res[2][2])))
    d = (res[0][1].double() - res[2][1]).abs()
    print("  worst ga entries", [(tuple(int(v) for v in torch.unravel_index(i, d.shape)), float(d.flatten()[i])) for i in d.flatten().topk(4).indices], float(res[2][1].abs().max()))
    for k in res[0][3]:
        print("  ", k, rel(res[0][3][k], res[2][3][k]), rel(res[1][3][k], res[2][3][k]))
