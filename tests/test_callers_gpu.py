"""GPU parity of the hot path's other callers: one training-mode forward_pass +
backward of DGL, RGL-NET (geometric data, 3 GNN iterations, loss summed over the iterations) and B-Global
(semantic data: Hungarian matching inside groups of identical parts, min-of-5 sampling, 32 noise channels) against
fixtures captured from the reference's own `build_model(cfg)` (tests/golden/make_golden.py).  Weights are rebuilt
from the parameter names by `param_fill` on both sides; the draws on the CPU generator (noise, randperm, GRU
initial state) are reproduced by seeding it identically."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import param_fill  # noqa: E402

from multi_part_assembly_amd import config  # noqa: E402
from multi_part_assembly_amd.pn_transformer import build_model  # noqa: E402

pytestmark = pytest.mark.gpu

CASES = {
    "dgl_step": config.dgl_everyday,
    "rgl_net_step": config.rgl_net_everyday,
    "global_semantic_step": config.global_partnet_chair,
    "pn_refine_step": config.pn_transformer_refine_everyday,
    # BASELINE.json configs[2] / configs[4]: the graph networks on the DGCNN encoder (cfg.model.encoder = 'dgcnn')
    "dgl_dgcnn_step": config.dgl_dgcnn_everyday,
    "rgl_net_dgcnn_artifact_step": config.rgl_net_dgcnn_artifact,
}
# Gradient tolerance.  The GNN callers stack 3 iterations of 512-wide BatchNorm + ReLU MLPs whose statistics come
# from 15-75 positions at the fixture's size, so the float32 REFERENCE itself is up to 4.4 % (DGL), 1.7 % (DGL + DGCNN,
# RGL-NET) away from the float64 evaluation of its own step (the `grad64.*` records of the fixtures; B-Global and the
# refine transformer: 1e-5).  The bar is therefore anchored at float64, per parameter tensor (errors relative to the
# tensor's largest float64 entry):
#   (a) |hip - fp64| <= 2 |ref_fp32 - fp64| + 1e-4, or
#   (b) |hip - fp64| <= GRAD_REL (1e-2 for the graph networks, 2e-3 else), or
#   (c) downstream of a ReLU flip that is SHOWN to be possible: the float64 evaluation of the reference has, in the SAME
#       MLP (`edge_mlps.0.`, `node_mlps.1.`, ...), a BatchNorm output — the pre-activation of a ReLU — within FLIP_PREACT
#       of zero relative to its channel's largest value (the `preact64.*` records of the fixtures, written by
#       make_golden.py's hooks on the float64 model: per module and channel the smallest and largest |output|).  A
#       float32 evaluation with another summation order can land on the other side of zero there; the unit's upstream
#       gradient then appears or vanishes in every parameter gradient of that MLP at and below the site (measured with
#       tools/debug_flip.py on RGL-NET + DGCNN: the site is edge_mlps.0.bn3 channel 1, |z64| = 3e-6 of its scale; its
#       own bias gradient moves 1.05e-2, conv3.weight's row 1 by 1.6e-2, conv2.weight by up to 1.26e-2 in 2 entries,
#       everything else of the MLP < 7e-3).  Under this clause a tensor may deviate by at most 5e-2, at most 2 of its
#       recorded entries by more than 1e-2 unless it is the site's own layer, and it must lie at or below (towards the
#       input of) a qualifying site of its own MLP — the number of excused tensors follows from the sites found, not
#       from a constant.  The test prints the site(s) and every tensor it excuses.
# A bias in front of a BatchNorm has a structurally zero gradient: there the bar is absolute, 1e-5 of the layer's
# weight-gradient scale (the float32 reference leaves 5e-7 there).
GRAD_REL = {"dgl_step": 1e-2, "rgl_net_step": 1e-2, "global_semantic_step": 2e-3, "pn_refine_step": 2e-3,
            "dgl_dgcnn_step": 1e-2, "rgl_net_dgcnn_artifact_step": 1e-2}
FLIP_PREACT = 5e-6  # |float64 pre-activation| / channel scale below which a float32 evaluation may land on the other side


@pytest.mark.parametrize("name", sorted(CASES))
def test_caller_step_matches_reference(golden, cuda_device, capsys, name):
    z = golden(name)
    cfg = CASES[name]()
    cfg.model.pc_feat_dim = int(z["cfg"][0])
    if name == "pn_refine_step":
        cfg.model.transformer_pos_enc = (64, 64)
        cfg.model.transformer_heads, cfg.model.transformer_feat_dim = int(z["cfg"][1]), int(z["cfg"][2])
    cfg.data.max_num_part = 5
    seed = int(z["seed"][0])
    torch.manual_seed(seed)
    model = build_model(cfg)
    assert sorted(model.state_dict().keys()) == [str(n) for n in z["names"]]  # same state_dict keys as upstream
    param_fill.fill_parameters(model, seed)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
    model.to(cuda_device).train()
    data = {k[5:]: torch.from_numpy(z[k].copy()).to(cuda_device) for k in z if k.startswith("data.")}
    torch.manual_seed(seed + 1)
    res = model.forward_pass(data, mode="train")
    res["loss"].backward()
    loss_err = {}
    for k in z:
        if k.startswith("loss."):
            loss_err[k[5:]] = abs(float(res[k[5:]]) - float(z[k])) / max(abs(float(z[k])), 1e-6)
    with capsys.disabled():
        worst_term = max(loss_err, key=loss_err.get)
        print(f"\n  {name}: {len(loss_err)} loss terms vs the reference: worst relative deviation {loss_err[worst_term]:.2e} "
              f"({worst_term})", end="")
    for k in z:
        if k.startswith("loss."):  # north_star's bar: 1e-4 relative, every term of every GNN iteration
            np.testing.assert_allclose(float(res[k[5:]]), float(z[k]), rtol=1e-4, atol=1e-6, err_msg=k)
    record = dict(z)
    rows, flips, used_sites = [], [], set()
    # ReLU sites where a float32 evaluation may flip: float64 pre-activations at the rounding level of zero
    sites = []
    for key, v in record.items():
        if key.startswith("preact64."):
            lo, hi = v
            sites += [(key[len("preact64."):], int(ch), float(lo[ch]), float(hi[ch])) for ch in np.nonzero(lo <= FLIP_PREACT * hi)[0]]
    for k, p in model.named_parameters():
        if ("grad." + k) in record or ("grad." + k + "#sample") in record:
            assert p.grad is not None, k
            g = p.grad.cpu().numpy()
            wscale = param_fill.grad64_scale(record, k[:-len("bias")] + "weight") if k.endswith(".bias") else 0.0
            if wscale > 0 and param_fill.grad64_scale(record, k) < 1e-9 * wscale:  # structurally zero gradient
                assert np.abs(g).max() <= 1e-5 * wscale, (k, float(np.abs(g).max()), wscale)
                continue
            mine, ref32, (outliers, n) = param_fill.anchored_errors(record, k, g, floor=1e-4)
            rows.append((mine, ref32, k))
            if mine <= 2.0 * ref32 + 1e-4 or mine <= GRAD_REL[name]:
                continue
            mlp = ".".join(k.split(".")[:2]) + "."
            near = [(m, ch, lo, hi) for (m, ch, lo, hi) in sites if m.startswith(mlp)]
            assert near, (k, mine, ref32, "no float64 pre-activation at the rounding level of zero in this MLP", sites)
            own_layer = any(m.rsplit(".", 1)[0] == k.rsplit(".", 2)[0] and m[-1] == k.rsplit(".", 1)[0][-1] for m, *_ in near)
            assert mine <= 5e-2 and (outliers <= 2 or own_layer), (k, mine, ref32, outliers, n, near)
            flips.append((k, f"{mine:.2e}", f"{outliers} of {n} entries > 1e-2"))
            used_sites.update(near)
    worst = max(rows)
    with capsys.disabled():
        med = sorted(r[0] for r in rows)[len(rows) // 2]
        print(f"\n  {name}: {len(rows)} gradient tensors vs float64: worst {worst[0]:.2e} ({worst[2]}; the float32 "
              f"reference there: {worst[1]:.2e}), median {med:.2e}; float32 reference worst {max(r[1] for r in rows):.2e}"
              f"; clause (c): {len(sites)} float64 pre-activations within {FLIP_PREACT:g} of zero in the model, used "
              f"(module, channel, |z64|, channel scale): {sorted(used_sites)}; tensors excused: {flips}", end="")
    # how many tensors clause (c) may excuse follows from the sites, not from a constant: a flip at layer L of an MLP
    # can only reach that MLP's parameter gradients at and below L (the unit's upstream gradient appears or vanishes on
    # its way back through layers L, L - 1, ...), so every excused tensor must be one of those — however many sites the
    # box at hand turns up
    for k, *_ in flips:
        mlp = ".".join(k.split(".")[:2]) + "."
        layer = int(k.rsplit(".", 2)[-2][-1])
        reach = max(int(m[-1]) for (m, ch, lo, hi) in sites if m.startswith(mlp))
        assert layer <= reach, (k, "upstream of every near-zero pre-activation of its MLP", [s_ for s_ in sites if s_[0].startswith(mlp)])
    for k, v in model.state_dict().items():
        if "running_" in k:
            param_fill.compare(record, "sd1.", k, v.cpu().numpy(), rel=1e-4)


@pytest.mark.parametrize("final_relu,rows,L,cin,feat", [(True, 7, 5, 128, 64), (False, 640, 1, 512, 128), (True, 40, 20, 256, 128),
                                                          (True, 131, 20, 256, 128), (False, 2049, 1, 128, 64)])
def test_pair_mlp_hip_layers_match_library_ops(cuda_device, final_relu, rows, L, cin, feat):
    """The Conv1d(k=1) + BatchNorm1d + ReLU layers of MLP3 / MLP4 / MLP5 on csrc/mlp.hip against the same module on
    torch's library ops (reference models/dgl/modules.py:5-58, rgl_net/modules.py:5-30): outputs, every gradient,
    running statistics; and evaluation mode.  Row counts on both sides of csrc/mlp.hip's switch between the
    one-tile-per-block GEMM (<= 2048 rows) and the 128-row-tile split-bf16 GEMM (2620 and 2049 rows here)."""
    import copy
    from multi_part_assembly_amd.gnn import _PairMLP
    torch.manual_seed(rows)
    mine = _PairMLP(cin, feat, final_relu=final_relu).to(cuda_device).train()
    mine.MIN_ROWS = 1  # force the HIP layers at these small sizes
    with torch.no_grad():
        for bn in (mine.bn1, mine.bn2, mine.bn3):
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_(0, 0.1)
    ref = copy.deepcopy(mine)
    ref64 = copy.deepcopy(mine).double()
    x = torch.randn(rows, L, cin, device=cuda_device)
    w = torch.randn(rows, L, feat, device=cuda_device)
    xa, xb, xc = x.clone().requires_grad_(), x.clone().requires_grad_(), x.double().requires_grad_()
    out = mine(xa)
    (out * w).sum().backward()
    want = ref._tail(ref.conv1(xb.transpose(1, 2)))
    (want * w).sum().backward()
    want64 = ref64._tail(ref64.conv1(xc.transpose(1, 2)))
    (want64 * w.double()).sum().backward()
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / (b.abs().max() + 1e-12))
    assert rel(out.detach(), want.detach()) < 1e-4
    assert rel(out.detach(), want64.detach()) < 1e-5

    def close(a, lib32, f64, who):
        """Three stacked BatchNorm backwards over few rows: anchored at the float64 evaluation of the same module — within
        2e-4 of it, or no further from it than twice the float32 library result is; a ReLU on the rounding edge may move
        isolated entries (at most 0.5 % of them beyond 2e-3, none beyond 3e-2)."""
        mine_e, lib_e = rel(a, f64), rel(lib32, f64)
        if mine_e < 2e-4 or mine_e <= 2.0 * lib_e:
            return
        far = ((a.double() - f64).abs() > 2e-3 * f64.abs().max()).float().mean()
        assert mine_e < 3e-2 and float(far) <= 0.005, (who, mine_e, lib_e, float(far))

    close(xa.grad, xb.grad, xc.grad, "x")
    for (k, p), (_, q), (_, r) in zip(mine.named_parameters(), ref.named_parameters(), ref64.named_parameters()):
        if "conv" in k and "bias" in k:   # a bias in front of a BatchNorm: its gradient is zero up to rounding
            assert float(p.grad.abs().max()) < 1e-3 * float(w.abs().sum())
        else:
            close(p.grad, q.grad, r.grad, k)
    for (k, a), (_, b) in zip(mine.named_buffers(), ref.named_buffers()):
        if "running" in k:
            assert rel(a, b) < 1e-4, k
    mine.eval()
    ref.eval()
    with torch.no_grad():
        assert rel(mine(x), ref._tail(ref.conv1(x.transpose(1, 2)))) < 1e-4


@pytest.mark.parametrize("B,P,F,same", [(3, 5, 64, False), (32, 20, 128, True), (2, 33, 128, False)])
def test_edge_mlp_first_layer_without_the_pair_tensor(cuda_device, B, P, F, same):
    """`_PairMLP.forward_pairs` with the first layer as two part-row GEMMs + a broadcast sum (csrc/mlp.hip:
    mpa_pair_layer_*) against (1) the same module with the first layer as a GEMM over the materialised [a_i ; b_j] rows
    and (2) the float64 library composition of the reference (dgl/network.py:135-152, dgl/modules.py:5-31): output, the
    gradients of both inputs (one tensor in both roles when `same`, as the networks call it) and of every parameter,
    running statistics and step counters."""
    import copy
    from multi_part_assembly_amd.gnn import _PairMLP
    torch.manual_seed(B * 100 + P)
    mine = _PairMLP(2 * F, 128).to(cuda_device).train()
    with torch.no_grad():
        for bn in (mine.bn1, mine.bn2, mine.bn3):
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_(0, 0.1)
    rows = copy.deepcopy(mine)
    rows.PAIR_LAYER = False
    ref64 = copy.deepcopy(mine).double()
    ref64.MIN_ROWS = 10 ** 12  # library ops
    a0 = torch.randn(B, P, F, device=cuda_device)
    b0 = a0 if same else torch.randn(B, P, F, device=cuda_device)
    w = torch.randn(B * P, P, 128, device=cuda_device)
    res = []
    for mod, cast in ((mine, torch.float32), (rows, torch.float32), (ref64, torch.float64)):
        a = a0.to(cast).clone().requires_grad_()
        b = a if same else b0.to(cast).clone().requires_grad_()
        out = mod.forward_pairs(a, b)
        (out * w.to(cast)).sum().backward()
        res.append((out.detach(), a.grad, None if same else b.grad, mod))
    rel = lambda x, y: float((x.double() - y.double()).abs().max() / (y.double().abs().max() + 1e-12))
    (o1, ga1, gb1, _), (o2, ga2, gb2, _), (o3, ga3, gb3, _) = res
    assert rel(o1, o3) < 1e-5 and rel(o2, o3) < 1e-5

    def close(x, x_rows, x64, who):
        """anchored at float64: within 2e-4 of it, or no further from it than twice the materialised-rows path is; a ReLU
        on the rounding edge (the two paths round the first layer differently: one 2F-term chain there, two F-term chains
        and a sum here) may move isolated entries — at most 0.5 % of them beyond 2e-3, none beyond 3e-2 (the rule of
        test_pair_mlp_hip_layers_match_library_ops)"""
        e, e_rows = rel(x, x64), rel(x_rows, x64)
        if e < 2e-4 or e <= 2.0 * e_rows:
            return
        far = ((x.double() - x64).abs() > 2e-3 * x64.abs().max()).float().mean()
        assert e < 3e-2 and float(far) <= 0.005, (who, e, e_rows, float(far))

    close(ga1, ga2, ga3, "a")
    if not same:
        close(gb1, gb2, gb3, "b")
    for (k, p1), (_, p2), (_, p3) in zip(mine.named_parameters(), rows.named_parameters(), ref64.named_parameters()):
        if "conv" in k and "bias" in k:  # a bias in front of a BatchNorm: zero up to rounding
            assert float(p1.grad.abs().max()) < 1e-3 * float(w.abs().sum())
        else:
            close(p1.grad, p2.grad, p3.grad, k)
    for (k, x), (_, y) in zip(mine.named_buffers(), ref64.named_buffers()):
        if "running" in k:
            assert rel(x, y) < 1e-4, k
        else:
            assert int(x) == int(y) == 1, k  # num_batches_tracked
    mine.eval()
    ref64.eval()
    with torch.no_grad():
        assert rel(mine.forward_pairs(a0, b0), ref64.forward_pairs(a0.double(), b0.double())) < 1e-4


def test_relation_net_hip_layers_match_library_ops(cuda_device):
    import copy
    from multi_part_assembly_amd.gnn import RelationNet
    torch.manual_seed(3)
    from multi_part_assembly_amd.gnn import _PairMLP
    mine = RelationNet().to(cuda_device)
    ref = copy.deepcopy(mine)
    x = torch.randn(8, 25 * 25, 256, device=cuda_device)  # 5000 pair rows: above _PairMLP.MIN_ROWS
    assert x.shape[0] * x.shape[1] >= _PairMLP.MIN_ROWS
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    out = mine(xa)
    out.square().sum().backward()
    want = torch.sigmoid(ref.mlp3(torch.relu(ref.mlp2(torch.relu(ref.mlp1(xb))))))
    want.square().sum().backward()
    rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-12))
    assert rel(out.detach(), want.detach()) < 1e-5
    assert rel(xa.grad, xb.grad) < 1e-4
    for (k, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        assert rel(p.grad, q.grad) < 1e-4, k


def test_relation_net_mask_is_the_callers_product(cuda_device):
    """`RelationNet(x, mask)` (mask folded into the head's launch) equals `RelationNet(x) * mask` on library ops, value and
    gradients (reference models/dgl/network.py:213: `relation_matrix * valid_matrix`)."""
    import copy
    from multi_part_assembly_amd.gnn import RelationNet
    torch.manual_seed(5)
    mine = RelationNet().to(cuda_device)
    ref = copy.deepcopy(mine).double()
    x = torch.randn(3, 20 * 20, 256, device=cuda_device)
    mask = (torch.rand(3, 20 * 20, device=cuda_device) < 0.6).float()
    xa, xb = x.clone().requires_grad_(), x.double().requires_grad_()
    w = torch.randn(3, 20 * 20, 1, device=cuda_device)
    out = mine(xa, mask)
    (out * w).sum().backward()
    want = torch.sigmoid(ref.mlp3(torch.relu(ref.mlp2(torch.relu(ref.mlp1(xb)))))) * mask.double()[..., None]
    (want * w.double()).sum().backward()
    rel = lambda a, b: float((a.double() - b).abs().max() / (b.abs().max() + 1e-12))
    assert out.shape == want.shape and rel(out.detach(), want.detach()) < 1e-5
    assert rel(xa.grad, xb.grad) < 1e-4
    for (k, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        assert rel(p.grad, q.grad) < 1e-4, k


@pytest.mark.parametrize("R,K,N", [(640, 7, 256), (5, 1, 3), (33, 16, 700)])
def test_narrow_linear_relu_matches_library_ops(cuda_device, R, K, N):
    """csrc/gnn_glue.hip's narrow layer (the pose encoder's 7 -> 256, reference models/dgl/modules.py:76-86) against
    float64 library ops: output, input / weight / bias gradients; without a bias; bit-reproducible."""
    from multi_part_assembly_amd.gnn_ops import narrow_linear_relu
    torch.manual_seed(R + N)
    lin = torch.nn.Linear(K, N).to(cuda_device)
    x = torch.randn(R, K, device=cuda_device)
    w = torch.randn(R, N, device=cuda_device)
    xa, xb = x.clone().requires_grad_(), x.double().requires_grad_()
    out = narrow_linear_relu(xa, lin.weight, lin.bias)
    (out * w).sum().backward()
    got = [xa.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone()]
    wd, bd = lin.weight.detach().double().requires_grad_(), lin.bias.detach().double().requires_grad_()
    want = torch.relu(xb @ wd.t() + bd)
    (want * w.double()).sum().backward()
    rel = lambda a, b: float((a.double() - b).abs().max() / (b.abs().max() + 1e-12))
    assert rel(out.detach(), want.detach()) < 1e-6
    for a, b in zip(got, (xb.grad, wd.grad, bd.grad)):
        assert rel(a, b) < 1e-5
    lin.zero_grad()
    xc = x.clone().requires_grad_()
    (narrow_linear_relu(xc, lin.weight, lin.bias) * w).sum().backward()
    assert torch.equal(xc.grad, got[0]) and torch.equal(lin.weight.grad, got[1]) and torch.equal(lin.bias.grad, got[2])
    nb = narrow_linear_relu(x.view(1, R, K), lin.weight)  # leading dimensions, no bias
    assert nb.shape == (1, R, N) and rel(nb, torch.relu(x.double() @ wd.detach().t())[None]) < 1e-6
    with pytest.raises(Exception):
        narrow_linear_relu(torch.randn(4, 17, device=cuda_device), torch.randn(8, 17, device=cuda_device))


@pytest.mark.parametrize("B,P,C", [(32, 20, 128), (2, 1, 5), (3, 64, 200)])
def test_relation_mean_matches_library_ops(cuda_device, B, P, C):
    """The relation-weighted mean of the edge features (reference models/dgl/network.py:135-152) in one launch against
    the float64 library formulation; rows without any relation weight (padded parts) give zeros as there."""
    from multi_part_assembly_amd.gnn_ops import relation_mean
    torch.manual_seed(B * P)
    edge = torch.randn(B, P, P, C, device=cuda_device)
    rel_w = torch.rand(B, P, P, device=cuda_device) * (torch.rand(B, P, P, device=cuda_device) < 0.7)
    rel_w[0, 0] = 0.0  # a padded part: no neighbours
    w = torch.randn(B, P, C, device=cuda_device)
    ea, ra = edge.clone().requires_grad_(), rel_w.clone().requires_grad_()
    eb, rb = edge.double().requires_grad_(), rel_w.double().requires_grad_()
    out = relation_mean(ea, ra)
    (out * w).sum().backward()
    want = (eb * rb[..., None]).sum(dim=2) / (rb.sum(dim=-1, keepdim=True) + 1e-6)
    (want * w.double()).sum().backward()
    rel = lambda a, b: float((a.double() - b).abs().max() / (b.abs().max() + 1e-12))
    assert rel(out.detach(), want.detach()) < 1e-5
    assert float(out[0, 0].abs().max()) == 0.0
    assert rel(ea.grad, eb.grad) < 1e-5
    assert rel(ra.grad, rb.grad) < 1e-4   # (entries of the empty row are O(1e6): the reference's 1 / 1e-6)
    out2 = relation_mean(edge.requires_grad_(), rel_w)  # weights that are data (the first iteration's valid matrix)
    (out2 * w).sum().backward()
    assert torch.equal(out2.detach(), out.detach()) and torch.equal(edge.grad, ea.grad)


@pytest.mark.parametrize("S,P,F,swap", [(32, 20, 128, False), (32, 20, 128, True), (2, 1, 4, False), (3, 7, 36, True)])
def test_pair_rows_match_library_ops(cuda_device, S, P, F, swap):
    """[part i ; part j] rows for every pair (reference models/dgl/network.py:121-125 — swapped halves — and 135-141):
    the same bytes as expand + cat, gradients = the sums over the other index."""
    from multi_part_assembly_amd.gnn_ops import pair_rows
    torch.manual_seed(S + P)
    a, b = torch.randn(S, P, F, device=cuda_device), torch.randn(S, P, F, device=cuda_device)
    w = torch.randn(S, P, P, 2 * F, device=cuda_device)
    aa, ba = a.clone().requires_grad_(), b.clone().requires_grad_()
    ad, bd = a.double().requires_grad_(), b.double().requires_grad_()
    out = pair_rows(aa, ba, swap=swap)
    (out * w).sum().backward()
    halves = [ad[:, :, None].expand(S, P, P, F), bd[:, None].expand(S, P, P, F)]
    want = torch.cat(halves[::-1] if swap else halves, dim=-1)
    (want * w.double()).sum().backward()
    assert torch.equal(out.detach().double(), want.detach())
    rel = lambda x, y: float((x.double() - y).abs().max() / (y.abs().max() + 1e-12))
    assert rel(aa.grad, ad.grad) < 1e-5 and rel(ba.grad, bd.grad) < 1e-5
    same = a.clone().requires_grad_()   # one tensor in both roles (the relation net's input): the two sums add up
    (pair_rows(same, same, swap=swap) * w).sum().backward()
    assert rel(same.grad, ad.grad + bd.grad) < 1e-5
    only_a = a.clone().requires_grad_()  # b is data
    (pair_rows(only_a, b, swap=swap) * w).sum().backward()
    assert torch.equal(only_a.grad, aa.grad)


def test_pose_encoder_hip_layers_match_library_ops(cuda_device):
    import copy
    from multi_part_assembly_amd.gnn import PoseEncoder
    torch.manual_seed(11)
    mine = PoseEncoder(7).to(cuda_device)
    ref = copy.deepcopy(mine).double()
    x = torch.randn(32, 20, 7, device=cuda_device)
    xa, xb = x.clone().requires_grad_(), x.double().requires_grad_()
    w = torch.randn(32, 20, 128, device=cuda_device)
    out = mine(xa)
    (out * w).sum().backward()
    want = torch.relu(ref.mlp2(torch.relu(ref.mlp1(xb))))
    (want * w.double()).sum().backward()
    rel = lambda a, b: float((a.double() - b).abs().max() / (b.abs().max() + 1e-12))
    assert out.shape == want.shape and rel(out.detach(), want.detach()) < 1e-5
    assert rel(xa.grad, xb.grad) < 1e-4
    for (k, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        assert rel(p.grad, q.grad) < 1e-4, k


@pytest.mark.parametrize("H,B,T", [(256, 32, 20), (128, 3, 5), (256, 7, 33)])
def test_gru_recurrent_matches_torch_gru(cuda_device, H, B, T):
    """csrc/gru.hip (all steps of both directions in one launch, tagged-word exchange per step) against torch.nn.GRU run per
    direction: hidden states of every step and the gradients of inputs, both weight sets and biases; and the masked
    bidirectional wrapper of RGL-NET (reference modules/rnn.py:6-46) with the HIP path against its library path."""
    import copy
    from multi_part_assembly_amd.gnn import _MaskedBiGRU
    torch.manual_seed(H + B)
    gru = torch.nn.GRU(input_size=H, hidden_size=H, num_layers=1, batch_first=True, bidirectional=True).to(cuda_device)
    mine = _MaskedBiGRU(gru)
    ref = _MaskedBiGRU(copy.deepcopy(gru))
    x = torch.randn(B, T, H, device=cuda_device)
    h0 = torch.randn(2, B, H, device=cuda_device)
    lengths = torch.randint(1, T + 1, (B,))
    lengths[0] = T
    valids = (torch.arange(T)[None] < lengths[:, None]).float().to(cuda_device)
    w = torch.randn(B, T, 2 * H, device=cuda_device)
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    out, _ = mine(xa, h0, valids=valids)
    (out * w).sum().backward()
    import multi_part_assembly_amd.gnn as G
    keep = G.gru_supported
    G.gru_supported = lambda *a: False  # library path of the same wrapper
    try:
        want, _ = ref(xb, h0, valids=valids)
        (want * w).sum().backward()
    finally:
        G.gru_supported = keep
    rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-12))
    assert rel(out.detach(), want.detach()) < 1e-5
    assert rel(xa.grad, xb.grad) < 1e-4
    for (k, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        assert rel(p.grad, q.grad) < 1e-4, k
    # bit-reproducible backward
    g1 = [p.grad.clone() for p in mine.parameters()]
    mine.zero_grad()
    out2, _ = mine(x.clone().requires_grad_(), h0, valids=valids)
    (out2 * w).sum().backward()
    assert all(torch.equal(a, p.grad) for a, p in zip(g1, mine.parameters()))


def test_gru_exchange_is_stable_over_many_launches(cuda_device):
    """The blocks of csrc/gru.hip hand each other every step's values as tagged 8-byte words that the consumers poll
    (no grid barrier): 60 forward + backward launches on the same inputs — also back to back on reused workspaces, where
    a stale word of the previous launch would carry a matching tag if the exchange buffers were not cleared — must
    give the same bits every time."""
    from multi_part_assembly_amd.gru import gru_recurrent, supported
    D, B, T, H = 2, 32, 20, 256
    if not supported(H, B, D):
        pytest.skip("device cannot hold the GRU grid")
    torch.manual_seed(5)
    gi = torch.randn(D, B, T, 3 * H, device=cuda_device)
    h0 = torch.randn(D, B, H, device=cuda_device)
    whh = (torch.randn(D, 3 * H, H, device=cuda_device) / H ** 0.5).requires_grad_()
    bhh = (0.1 * torch.randn(D, 3 * H, device=cuda_device)).requires_grad_()
    w = torch.randn(D, B, T, H, device=cuda_device)
    first = None
    for _ in range(60):
        g = gi.clone().requires_grad_()
        out = gru_recurrent(g, h0, whh, bhh)
        grads = torch.autograd.grad((out * w).sum(), [g, whh, bhh])
        got = [out.detach()] + [t.detach() for t in grads]
        if first is None:
            first = [t.clone() for t in got]
            assert all(torch.isfinite(t).all() for t in first)
        else:
            assert all(torch.equal(a, b) for a, b in zip(first, got))


def _against_float64(cuda_device, capsys, cfg, label, oracle_loss, B=4, no_dropout=False, slack=0.17, median_cap=1.2,
                     abs_cap=None, preset="everyday", reseed=None):
    """One training-mode forward_pass + backward at P = 20, N = 1000 on the HIP path, on the float32 CPU oracle and on the
    same oracle in float64; returns nothing, asserts that the HIP path is as close to float64 as the float32 oracle."""
    from multi_part_assembly_amd import synthetic
    import statistics
    torch.manual_seed(0)
    model = build_model(cfg)
    if no_dropout:
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
            if isinstance(m, torch.nn.MultiheadAttention):
                m.dropout = 0.0
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    names = [k for k, _ in model.named_parameters()]
    batch = synthetic.make_batch(B, 20, 1000, preset=preset, seed=1234, device=cuda_device)
    batch.pop("num_parts", None)
    model.to(cuda_device).train()
    if reseed is not None:  # models that draw on the CPU generator inside forward: the same draws for all three evaluations
        torch.manual_seed(reseed)
    loss = model.training_step(batch, 0)
    loss.backward()
    hip = {k: p.grad.detach().cpu().double() for k, p in model.named_parameters() if p.grad is not None}
    hip_loss = float(loss.detach())
    threads = torch.get_num_threads()
    torch.set_num_threads(16)

    def oracle(dt):
        sd = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
        params = {k: sd[k].requires_grad_() for k in names}
        cb = {k: (v.cpu().to(dt) if v.is_floating_point() else v.cpu()) for k, v in batch.items() if hasattr(v, "cpu")}
        if reseed is not None:
            torch.manual_seed(reseed)
        total = oracle_loss(sd, cb)
        total.backward()
        return float(total.detach()), {k: p.grad.double() for k, p in params.items() if p.grad is not None}

    try:
        l32, g32 = oracle(torch.float32)
        l64, g64 = oracle(torch.float64)
    finally:
        torch.set_num_threads(threads)
    assert abs(hip_loss - l64) <= 2.0 * abs(l32 - l64) + 1e-4 * abs(l64), (hip_loss, l32, l64)
    ratios, worst, worst_o, needed = [], (0.0, ""), 0.0, []
    for k, b in g64.items():
        scale = float(b.abs().max())
        if scale < 1e-10:  # a bias in front of a BatchNorm: structurally zero
            continue
        eh = float((hip[k] - b).abs().max()) / scale
        eo = float((g32[k] - b).abs().max()) / scale
        assert eh <= 2.0 * eo + slack, (k, eh, eo)
        assert abs_cap is None or eh <= abs_cap, (k, eh)
        if eh > 2.0 * eo:
            needed.append((eh - 2.0 * eo, k))
        ratios.append(eh / max(eo, 1e-12))
        worst = max(worst, (eh, k))
        worst_o = max(worst_o, eo)
    med = statistics.median(ratios)
    with capsys.disabled():
        print(f"\n  {label} at P=20, N=1000, B={B}: loss hip {hip_loss:.6f} / oracle32 {l32:.6f} / float64 {l64:.6f}; "
              f"gradient deviation from float64, hip : oracle32, median ratio over {len(ratios)} tensors {med:.2f} "
              f"(largest deviation of a tensor: hip {worst[0]:.2e} [{worst[1]}], oracle32 {worst_o:.2e}); tensors beyond "
              f"2 x the oracle's deviation (the `slack` term of the bar, {slack:g}): {len(needed)} of {len(ratios)}"
              + (f", largest excess {max(needed)[0]:.2e} [{max(needed)[1]}]" if needed else ""), end="")
    assert median_cap is None or med <= median_cap, med


def test_dgl_dgcnn_step_at_the_benchmark_part_size_against_float64(cuda_device, capsys):
    """BASELINE.json configs[2] at its part size (P = 20 slots, N = 1000 points, everyday-like synthetic shapes; B = 4
    to bound the CPU time): the training-mode forward_pass + backward of DGL + DGCNN on the HIP path, on the float32 CPU
    oracle (oracle/callers.py) and on the same oracle in float64.

    At this size the step is NOT well conditioned in float32: the loss goes through ~10^5 discrete choices per sample
    (nearest neighbours of the Chamfer terms, kNN graphs, max-pool / ReLU selections) on random-init predictions, and the
    float32 oracle's own parameter gradients sit 5-80 % (largest entry, per tensor) from its float64 evaluation.  A
    bar "hip == oracle32 to 1e-2" would therefore fail for ANY correct float32 implementation; the bar here is that the
    HIP path is as close to float64 as the float32 restatement of the reference is: the loss within twice the
    oracle's own deviation, per tensor |hip - f64| <= 2 |o32 - f64| + 0.17, and the median ratio of the two deviations
    over all tensors <= 1.2 (measured: 0.88 — the HIP path is the closer of the two).  The additive term is the measured
    need + 50 %: 3 of the 105 tensors sit beyond twice the oracle's deviation, the farthest by 0.113
    (pose_predictors.1.fc_layers.0.weight); the test prints both numbers on every run."""
    from oracle import callers as oc
    cfg = config.dgl_dgcnn_everyday()
    _against_float64(cuda_device, capsys, cfg, "DGL + DGCNN",
                     lambda sd, cb: oc.dgl_loss(sd, cb, cfg.model.gnn_iter, cfg.model.encoder, True, {})["loss"])


def test_rgl_net_dgcnn_step_at_the_benchmark_part_size_against_float64(cuda_device, capsys):
    """BASELINE.json configs[4] at its part size (RGL-NET + DGCNN, artifact-like clouds: 12-20 small parts of 1000 points;
    B = 4): the same float64-anchored bar as the DGL step above, against `oracle/callers.py` with the recurrent node update
    (pinned by both RGL-NET reference fixtures in tests/test_oracle_golden.py).  All three evaluations see the same GRU
    initial states (the CPU generator is re-seeded in front of each).  Measured: loss 2.702235 (HIP) / 2.704128 (float32
    oracle) / 2.703485 (float64); median ratio of the per-tensor gradient deviations 0.59 over 135 tensors."""
    from oracle import callers as oc
    cfg = config.rgl_net_dgcnn_artifact()
    _against_float64(cuda_device, capsys, cfg, "RGL-NET + DGCNN",
                     lambda sd, cb: oc.dgl_loss(sd, cb, cfg.model.gnn_iter, cfg.model.encoder, True, {}, recurrent=True,
                                                merge_node=cfg.model.merge_node)["loss"],
                     preset="artifact", reseed=4321, slack=0.21)  # measured need 0.136 (8 of 135 tensors) + 50 %


def test_pn_transformer_step_at_the_benchmark_part_size_against_float64(cuda_device, capsys):
    """BASELINE.json configs[1] — the configuration the headline metric is quoted on (PNTransformer + PointNet, P = 20,
    N = 1000; B = 4, dropout off so that all three evaluations see the same network).  This step IS well conditioned (no
    kNN graph, one max-pool): the float32 oracle sits 2e-5 .. 7e-4 from float64 and the bar is absolute — every parameter
    gradient of the HIP path within 1e-3 of float64 (largest entry of the tensor; measured 1.2e-4 .. 8.1e-4, the largest on
    `encoder.conv4.weight` where the float32 oracle is at 7.4e-4), and within 2 x the oracle's deviation + 4e-4 (measured
    need 2.6e-4 on encoder.conv5.weight, + 50 %; 61 of 73 tensors sit beyond twice the oracle's tiny deviation).  The
    transformer and pose-head gradients share a ~2e-4 offset that enters with d loss / d rot at the pose head's output (2e-4
    there for the HIP path, 2e-5 for the float32 oracle; the translation gradient: 3e-5 for both).  It is not arithmetic:
    fed the SAME predicted poses, the fused loss backward is within 1e-7 of float64 per term, like the oracle — except the
    per-part Chamfer term, 6e-5 for BOTH float32 evaluations (nearest-neighbour near-ties resolve differently in float64).
    The predicted poses themselves differ from float64 by 3e-6 (HIP) / 2e-6 (oracle), and which near-ties those shifts
    flip decides the offset; evaluating the pose head's normalisation backward in double changes nothing.
    A/B, round 5: is it the fp32-grade split-bf16 products (the 128 / 256-wide PointNet layers)?  The same test against a
    build with every PointNet GEMM on `v_mfma_f32_32x32x2_f32` (tools/build_variant.sh pn_exact pointnet.hip
    -DMPA_PN_SPLIT=0; the transformer and the pose head are exact-fp32 matrix products in both builds): median ratio
    5.49 (split: 6.70), 64 of 73 tensors beyond twice the oracle's deviation (66), largest 8.00e-4 (8.02e-4) on the same
    tensor.  The split products account for a fifth of the ratio; the offset is there without them."""
    from oracle import nets as on
    cfg = config.pn_transformer_everyday()
    _against_float64(cuda_device, capsys, cfg, "PNTransformer + PointNet",
                     lambda sd, cb: on.pn_transformer_loss(sd, cb, cfg.model.transformer_layers,
                                                           cfg.model.transformer_heads, training=True, stats_out={})[0]["loss"],
                     no_dropout=True, slack=4e-4, median_cap=None, abs_cap=1e-3)


def test_dgl_dgcnn_graph_replay_equals_eager_steps(cuda_device):
    """BASELINE.json configs[2] as ONE HIP graph (bench.py --config c3 --graph): DGL draws no random numbers in
    training, every kernel on its path is deterministic (no atomics: fixed-order statistics, the transposed kNN graph
    in the encoder's backward), and the device-side valid-part count keeps the launch shapes static — so the replayed
    graph must walk the eager trajectory to the last bit, with a different batch (other part counts) every step."""
    from multi_part_assembly_amd import synthetic
    from multi_part_assembly_amd.trainer import Trainer

    def fresh(use_graph):
        cfg = config.dgl_dgcnn_everyday()
        torch.manual_seed(3)
        model = build_model(cfg).to(cuda_device)
        return Trainer(model, cfg, use_graph=use_graph, graph_warmup=1)

    eager, graph = fresh(False), fresh(True)
    for step in range(4):
        batch = synthetic.make_batch(3, 20, 256, preset="everyday", seed=50 + step, device=cuda_device)
        batch.pop("num_parts")
        le = eager.train_step(dict(batch))
        lg = graph.train_step(dict(batch))
        assert float(lg) == float(le), (step, float(lg), float(le))
    assert graph._graph is not None
    assert torch.equal(graph.flat.flat_param, eager.flat.flat_param)


def test_rgl_net_step_captures_as_one_graph(cuda_device):
    """BASELINE.json configs[4] in graph mode (bench.py --config c5 --graph): the GRU's random initial state is the
    only host-side draw on RGL-NET's path; under capture it comes from the device generator (gnn.py), so the step has
    no host-to-device copy left and replays with fresh noise — the replays must train (finite, moving loss) and draw
    different initial states each time."""
    from multi_part_assembly_amd import synthetic
    from multi_part_assembly_amd.trainer import Trainer

    cfg = config.rgl_net_dgcnn_artifact()
    torch.manual_seed(5)
    model = build_model(cfg).to(cuda_device)
    tr = Trainer(model, cfg, use_graph=True, graph_warmup=1)
    batch = synthetic.make_batch(3, 20, 256, preset="artifact", seed=9, device=cuda_device)
    batch.pop("num_parts")
    losses = [float(tr.train_step(dict(batch))) for _ in range(5)]
    assert tr._graph is not None
    assert all(np.isfinite(losses)), losses
    assert len(set(losses[1:])) == 4, losses  # same batch, new noise + updated weights: no replay repeats a value


# ---- HIP-graph replay of the graph networks (round-3 advisor finding: the GRU's tagged exchange words) -------------------
def _graph_net_trainer(golden, cuda_device, name, **kw):
    from multi_part_assembly_amd.trainer import Trainer
    z = golden(name)
    cfg = CASES[name]()
    cfg.model.pc_feat_dim = int(z["cfg"][0])
    cfg.data.max_num_part = 5
    cfg.optimizer.lr_scheduler = ""
    seed = int(z["seed"][0])
    torch.manual_seed(seed)
    model = build_model(cfg)
    param_fill.fill_parameters(model, seed)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.to(cuda_device)
    data = {k[5:]: torch.from_numpy(z[k].copy()).to(cuda_device) for k in z if k.startswith("data.")}
    return Trainer(model, cfg, **kw), data


@pytest.mark.parametrize("name", ["dgl_step", "rgl_net_step", "rgl_net_dgcnn_artifact_step"])
def test_graph_replay_equals_eager_steps_of_graph_networks(golden, cuda_device, monkeypatch, name):
    """Captured-graph replays of DGL / RGL-NET walk the eager trajectory.  RGL-NET's recurrent kernels hand hidden states
    between blocks as {value, tag} words with tags 1..T in EVERY launch: unless the words are cleared by a graph-safe
    node before each replay, a consumer can take the PREVIOUS replay's word for the current one.  The GRU's random
    initial state is pinned to one fixed device tensor on both sides (eager draws it on the CPU generator, capture on
    the device generator: two different streams by design), everything else is the shipped step."""
    from multi_part_assembly_amd import gnn
    fixed = {}

    def fixed_hidden(self, B, device=None):
        key = (B, self.pc_feat_dim)
        if key not in fixed:
            g = torch.Generator().manual_seed(11)
            fixed[key] = torch.randn((2, B, 2 * self.pc_feat_dim), generator=g).to(device)
        return fixed[key]

    if hasattr(gnn, "RGLNet"):
        monkeypatch.setattr(gnn.RGLNet, "_init_gru_hidden", fixed_hidden)
    eager, batch = _graph_net_trainer(golden, cuda_device, name)
    graph, _ = _graph_net_trainer(golden, cuda_device, name, use_graph=True, graph_warmup=1)
    for step in range(6):
        le = eager.train_step(batch)
        lg = graph.train_step(batch)
        assert float(lg) == float(le), (step, float(lg), float(le))
    assert graph._graph is not None
    assert torch.equal(graph.flat.flat_param, eager.flat.flat_param), \
        float((graph.flat.flat_param - eager.flat.flat_param).abs().max())


def test_graph_mode_falls_back_to_eager_for_semantic_matching(cuda_device):
    """B-Global (BASELINE.json configs[0]) matches identical parts on a point sample drawn on the host every step
    (base_model.py:196-238 of the reference): a captured step cannot carry that.  Trainer(use_graph=True) says so and runs
    eager launches instead of failing inside the capture; the steps train as the eager trainer's do."""
    import warnings

    import bench
    from multi_part_assembly_amd.trainer import Trainer
    cfg, batch, _, B, P = bench.workload("c1", 0, cuda_device)
    batch.pop("num_parts", None)
    torch.manual_seed(0)
    model = build_model(cfg).to(cuda_device)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        tr = Trainer(model, cfg, use_graph=True)
    assert not tr.use_graph and any("semantic part matching" in str(w.message) for w in caught)
    losses = [float(tr.train_step(batch, i)) for i in range(4)]
    assert all(np.isfinite(losses))
