"""Model-level parity on the GPU: PNTransformer step vs the fixture captured from the reference
(forward_pass + backward on the same weights and batch), encoders / transformer vs their fixtures,
fused Adam vs torch.optim.Adam, and the Trainer step vs the CPU oracle."""
import numpy as np
from pathlib import Path
import pytest
import torch

from multi_part_assembly_amd import config
from multi_part_assembly_amd.encoder import build_encoder
from multi_part_assembly_amd.optim import FusedAdam
from multi_part_assembly_amd.pn_transformer import build_model
from multi_part_assembly_amd.regressor import StocasticPoseRegressor
from multi_part_assembly_amd.trainer import Trainer
from multi_part_assembly_amd.transformer import TransformerEncoder

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _load(module, z, prefix):
    sd = {k[len(prefix):]: T(v.copy()) for k, v in z.items() if k.startswith(prefix)}
    missing = module.load_state_dict(sd, strict=True)
    return missing


def _no_dropout(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0


def _rel(a, b):
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-12)


@pytest.mark.parametrize("name,feat", [("pointnet", 256), ("dgcnn", 128)])
def test_encoder_matches_reference(golden, cuda_device, name, feat):
    z = golden(name)
    enc = build_encoder(name, feat)
    _load(enc, z, "sd0.")
    enc.to(cuda_device).train()
    x = T(z["x"]).to(cuda_device).requires_grad_()
    out = enc(x)
    (out * T(z["w"]).to(cuda_device)).sum().backward()
    assert _rel(out.detach().cpu().numpy(), z["feat_train"]) < 1e-4
    # DGCNN: the fixture's float32 reference gradients are THEMSELVES 2.8e-2 (grad_x) and 2.4e-3 (conv1.weight) away from
    # the float64 evaluation of the same network on the same graphs — one near-tie of the max over the 20 neighbours
    # resolved the other way in the reference's float32 run — while this build is within 5e-6 of float64 on every tensor
    # (tests/test_dgcnn_gpu.py::test_encoder_gradients_with_reference_graphs).  So those two tensors are compared with
    # the reference's own deviation as the bar, every other tensor at 2e-4.
    ref_dev = {"grad_x": 3e-2, "conv1.0.weight": 3e-3}

    def close(a, b, who):
        tol = 1e-3 if name == "pointnet" else ref_dev.get(who, 2e-4)
        assert _rel(a, b) < tol, (who, _rel(a, b))
        if name == "dgcnn" and who in ref_dev:  # ... and the deviation is isolated: the bulk of the entries agree
            bad = (np.abs(a - b) > 2e-4 * np.abs(b).max()).mean()
            assert bad < 0.05, (who, float(bad))

    if name == "dgcnn":  # the HIP PointNet does not differentiate w.r.t. its input points (data)
        close(x.grad.cpu().numpy(), z["grad_x"], "grad_x")
    for k, p in enc.named_parameters():
        close(p.grad.cpu().numpy(), z["grad." + k], k)
    for k, v in enc.state_dict().items():  # running statistics after the training-mode forward
        np.testing.assert_allclose(v.cpu().numpy(), z["sd1." + k], rtol=1e-4, atol=1e-5, err_msg=k)
    enc.eval()
    with torch.no_grad():
        assert _rel(enc(x).cpu().numpy(), z["feat_eval"]) < 1e-4


@pytest.mark.parametrize("C", [3, 64, 128])
def test_knn_kernel_selects_the_k_best(golden, cuda_device, C):
    """mpa_knn_exact against fp64 scores of the reference's formula (dgcnn.py:8-15): 20 distinct neighbours per point,
    best first, self included, and none of them worse than the true 20th best by more than rounding — a property
    check that does not go through the C oracle (tests/test_dgcnn_gpu.py holds the index-exact comparisons)."""
    from multi_part_assembly_amd.encoder import knn_exact
    g = torch.Generator().manual_seed(C)
    n, N, k = 3, 300, 20
    x = torch.randn(n, N, C, generator=g) * (0.3 if C == 3 else 1.0)
    rows = x.reshape(n * N, C)
    if C == 3:
        rows = torch.cat([rows, torch.zeros(n * N, 1)], dim=1)
    idx = knn_exact(rows.to(cuda_device).contiguous(), n, N, C).cpu().view(n, N, k).long()
    xd = x.double()
    score = -((xd[:, :, None, :] - xd[:, None, :, :]) ** 2).sum(-1)            # [n, N, N]
    mine = torch.gather(score, 2, idx)
    kth = score.topk(k, dim=-1)[0][..., -1:]
    tol = 1e-5 * (xd ** 2).sum(-1).max()
    assert (mine >= kth - tol).all()
    assert (mine[..., :-1] >= mine[..., 1:] - tol).all()                        # best first
    assert (idx.sort(-1)[0].diff(dim=-1) > 0).all()                            # distinct
    assert (idx[..., 0] == torch.arange(N)[None]).all()                        # the point itself comes first


def test_encoders_outside_the_kernels_instantiation_take_the_library_path(cuda_device):
    """One policy for every module (encoder.py, transformer.py): what the hand-written kernels are not instantiated for —
    but the reference accepts (encoder/pointnet.py:6-41, dgcnn.py:41-109: any feat_dim, per-point features, any number of
    points) — runs on library operators, with one warning per module.  (a) The library path IS the module's function:
    forced on an in-envelope configuration (`force_library`) it reproduces the HIP path's features, parameter gradients
    and running statistics, masked parts included.  (b) Out-of-envelope configurations run and have the reference's
    output shapes.  (c) Fewer than 20 points per DGCNN cloud fail as upstream (no 20-neighbour graph exists)."""
    import warnings

    torch.manual_seed(5)
    for arch, N in (("pointnet", 300), ("dgcnn", 96)):
        a, b = build_encoder(arch, 128).to(cuda_device), build_encoder(arch, 128).to(cuda_device)
        b.load_state_dict(a.state_dict())
        b.force_library = True
        x = (torch.randn(5, N, 3) * 0.3).to(cuda_device)
        valid = torch.tensor([1, 0, 1, 1, 0.0], device=cuda_device)
        w = torch.randn(5, 128, device=cuda_device)
        fa = a.forward_parts(x, valid)
        (fa * w).sum().backward()
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            fb = b.forward_parts(x, valid)
            fb2 = b.forward_parts(x, valid)  # (one warning per module, not per call)
        assert sum("library operators" in str(r.message) for r in rec) == 1
        (fb * w).sum().backward()
        assert torch.equal(fb[valid == 0], torch.zeros_like(fb[valid == 0]))
        np.testing.assert_allclose(fa.detach().cpu().numpy(), fb.detach().cpu().numpy(), rtol=2e-4, atol=2e-5, err_msg=arch)
        # The library path's backward scatters with float atomics (index / gather gradients of PyTorch-ROCm): its own
        # gradients move in the last bits from run to run, and a max over 20 neighbours that sits on a rounding edge then
        # routes a gradient elsewhere (seen once in ~25 runs of this test, bn2.bias of the DGCNN: 5e-3 exceeded).  The HIP
        # path is bit-reproducible (tests/test_fuzz_gpu.py `repro`): the library side is re-evaluated, at most twice more.
        for attempt in range(3):
            worst = max((_rel(p.grad.cpu().numpy(), q.grad.cpu().numpy()), k)
                        for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()))
            if worst[0] < 5e-3:
                break
            for q in b.parameters():
                q.grad = None
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                (b.forward_parts(x, valid) * w).sum().backward()
        assert worst[0] < 5e-3, (arch, worst)
        del fb2
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        x = torch.randn(3, 40, 3, device=cuda_device)
        assert build_encoder("pointnet", 96).to(cuda_device)(x).shape == (3, 96)
        assert build_encoder("pointnet", 128, global_feat=False).to(cuda_device)(x).shape == (3, 40, 128)
        assert build_encoder("dgcnn", 80).to(cuda_device)(x).shape == (3, 80)
        assert build_encoder("dgcnn", 128, global_feat=False).to(cuda_device)(x).shape == (3, 40, 128)
        big = build_encoder("dgcnn", 128).to(cuda_device)
        assert big(torch.randn(1, 1025, 3, device=cuda_device)).shape == (1, 128)  # beyond the fused kernels' 1024 points
        with pytest.raises(RuntimeError, match="at least 20 points"):
            big(torch.zeros(2, 19, 3, device=cuda_device))


def test_pointnet_masked_parts_equal_compacted(cuda_device):
    """forward_parts(all slots, mask) == forward(valid parts only): features, gradients, running stats."""
    torch.manual_seed(3)
    a, b = build_encoder("pointnet", 128).to(cuda_device), build_encoder("pointnet", 128).to(cuda_device)
    b.load_state_dict(a.state_dict())
    x = (torch.randn(7, 300, 3) * 0.3).to(cuda_device)   # N not a multiple of the 64-row wave tile
    valid = torch.tensor([1, 0, 1, 1, 0, 0, 1.0], device=cuda_device)
    w = torch.randn(7, 128, device=cuda_device)
    fa = a.forward_parts(x, valid)
    (fa * w).sum().backward()
    keep = valid.bool()
    fb = b(x[keep])
    (fb * w[keep]).sum().backward()
    assert torch.equal(fa[~keep], torch.zeros_like(fa[~keep]))
    np.testing.assert_allclose(fa[keep].detach().cpu().numpy(), fb.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
    for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert _rel(p.grad.cpu().numpy(), q.grad.cpu().numpy()) < 1e-4, k
    for (k, u), (_, v) in zip(a.state_dict().items(), b.state_dict().items()):
        np.testing.assert_allclose(u.cpu().numpy(), v.cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg=k)


@pytest.mark.parametrize("mixed_gamma", [False, True])
def test_pointnet_matches_torch_ops_at_full_width(cuda_device, mixed_gamma):
    """N=1000 (4 row tiles per part, ragged tail), F=256: against the same network written with stock
    torch ops on the GPU (Conv1d/BatchNorm1d), forward and all parameter gradients.  mixed_gamma: the last
    BatchNorm has negative and zero weights too (its max over points then is a min / a constant of the
    pre-BatchNorm values, which the HIP path never stores)."""
    import torch.nn.functional as Fn

    torch.manual_seed(4)
    enc = build_encoder("pointnet", 256).to(cuda_device).train()
    if mixed_gamma:
        with torch.no_grad():
            enc.bn5.weight[::3] *= -1.0
            enc.bn5.weight[1::7] = 0.0
            enc.bn5.bias.normal_()
            enc.bn4.weight[::5] *= -0.5
    x = (torch.randn(9, 1000, 3) * 0.2).to(cuda_device)
    w = torch.randn(9, 256, device=cuda_device)
    ref_params = {k: v.detach().clone().requires_grad_() for k, v in enc.named_parameters()}
    h = x.transpose(2, 1)
    for i in range(1, 6):
        h = Fn.conv1d(h, ref_params[f"conv{i}.weight"])
        h = Fn.batch_norm(h, None, None, ref_params[f"bn{i}.weight"], ref_params[f"bn{i}.bias"], True, 0.1, 1e-5)
        if i < 5:
            h = Fn.relu(h)
    ref = h.max(dim=-1)[0]
    (ref * w).sum().backward()
    out = enc(x)
    (out * w).sum().backward()
    assert _rel(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) < 1e-4
    for k, p in enc.named_parameters():
        assert _rel(p.grad.cpu().numpy(), ref_params[k].grad.cpu().numpy()) < 1e-3, k


def test_transformer_and_pose_head_match_reference(golden, cuda_device):
    z = golden("transformer")
    d, heads, ffn, layers = (int(v) for v in z["cfg"])
    enc = TransformerEncoder(d, heads, ffn, layers, norm_first=True, dropout=0.0)
    head = StocasticPoseRegressor(feat_dim=d, noise_dim=0)
    _load(enc, z, "enc.")
    _load(head, z, "head.")
    enc.to(cuda_device).train()
    head.to(cuda_device).train()
    tok = T(z["tokens"]).to(cuda_device).requires_grad_()
    valid = T(z["valid"]).to(cuda_device)
    feats = enc(tok, valid)
    rot, trans = head(feats)
    vm = valid[..., None].float()
    ((rot * T(z["w_rot"]).to(cuda_device) * vm).sum() + (trans * T(z["w_trans"]).to(cuda_device) * vm).sum()).backward()
    v = z["valid"]
    assert _rel(feats.detach().cpu().numpy()[v], z["feats"][v]) < 1e-4
    assert _rel(rot.detach().cpu().numpy()[v], z["rot"][v]) < 1e-4
    assert _rel(trans.detach().cpu().numpy()[v], z["trans"][v]) < 1e-4
    for k, p in enc.named_parameters():
        assert _rel(p.grad.cpu().numpy(), z["genc." + k]) < 1e-3, k
    for k, p in head.named_parameters():
        assert _rel(p.grad.cpu().numpy(), z["ghead." + k]) < 1e-3, k


def test_transformer_dropout_matches_oracle_with_same_masks(golden, cuda_device):
    """Training-mode dropout (p = 0.1 upstream, transformer.py:10): the oracle regenerates the HIP kernels'
    counter-based masks (oracle.nets.dropout_keep_scale), so outputs and every gradient are compared value by
    value — forward masks, the masks regenerated in backward and the ReLU/dropout gradient gates included."""
    from multi_part_assembly_amd.transformer import _TransformerFn
    from oracle import nets as on
    z = golden("transformer")
    d, heads, ffn, layers = (int(v) for v in z["cfg"])
    enc = TransformerEncoder(d, heads, ffn, layers, norm_first=True, dropout=0.1)
    _load(enc, z, "enc.")
    p_drop, seed = 0.1, 0xC0FFEE1234567
    sd = {k: v.detach().clone().requires_grad_() for k, v in enc.state_dict().items()}
    tok_ref = T(z["tokens"]).requires_grad_()
    valid = T(z["valid"])
    w = torch.randn(*z["tokens"].shape, generator=torch.Generator().manual_seed(5))
    ref = on.transformer_encoder(tok_ref, valid, sd, "", layers, heads, dropout_p=p_drop, seed=seed)
    (ref * w).sum().backward()
    assert not torch.allclose(ref, on.transformer_encoder(tok_ref, valid, sd, "", layers, heads))  # masks bite
    enc.to(cuda_device).train()
    tok = T(z["tokens"]).to(cuda_device).requires_grad_()
    out = _TransformerFn.apply(tok, valid.reshape(-1).float().to(cuda_device), heads, p_drop, seed, None, *enc._params())
    (out * w.to(cuda_device)).sum().backward()
    assert _rel(out.detach().cpu().numpy(), ref.detach().numpy()) < 1e-4
    assert _rel(tok.grad.cpu().numpy(), tok_ref.grad.numpy()) < 1e-3
    for k, p in enc.named_parameters():
        assert _rel(p.grad.cpu().numpy(), sd[k].grad.numpy()) < 1e-3, k
    # module-level behaviour: a fresh mask per call in train mode, none in eval mode
    with torch.no_grad():
        a, b = enc(tok, valid.to(cuda_device)), enc(tok, valid.to(cuda_device))
        assert not torch.equal(a, b)
        enc.eval()
        a, b = enc(tok, valid.to(cuda_device)), enc(tok, valid.to(cuda_device))
        assert torch.equal(a, b)
        clean = on.transformer_encoder(T(z["tokens"]), valid, {k: v.detach() for k, v in sd.items()}, "", layers, heads)
        assert _rel(a.cpu().numpy(), clean.numpy()) < 1e-4


@pytest.mark.parametrize("B,P", [(3, 7), (5, 14)])
def test_transformer_width_256_dropout_matches_oracle_with_same_masks(cuda_device, B, P):
    """D = 256 is the width at which both LayerNorm passes ride in GEMM operand loads (tf_gemm.h LNM = 1 / 2): forward,
    the d x / masked d x / dgamma / dbeta of every fused LayerNorm backward and the ragged last 32-row tile (M = 21, 70)
    against the oracle regenerating the same counter-based masks."""
    from multi_part_assembly_amd.transformer import _TransformerFn
    from oracle import nets as on
    torch.manual_seed(11)
    D, H, FF, L = 256, 8, 1024, 3
    enc = TransformerEncoder(D, H, FF, L, norm_first=True, dropout=0.1)
    with torch.no_grad():
        for p in enc.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    p_drop, seed = 0.1, 0xFEED5EED77
    sd = {k: v.detach().clone().requires_grad_() for k, v in enc.state_dict().items()}
    g = torch.Generator().manual_seed(3)
    num = torch.randint(2, P + 1, (B,), generator=g)
    valid = torch.arange(P)[None] < num[:, None]
    tok0 = torch.randn(B, P, D, generator=g) * valid[..., None]
    w = torch.randn(B, P, D, generator=g)
    tok_ref = tok0.clone().requires_grad_()
    ref = on.transformer_encoder(tok_ref, valid, sd, "", L, H, dropout_p=p_drop, seed=seed)
    (ref * w).sum().backward()
    enc.to(cuda_device).train()
    tok = tok0.to(cuda_device).requires_grad_()
    out = _TransformerFn.apply(tok, valid.reshape(-1).float().to(cuda_device), H, p_drop, seed, None, *enc._params())
    (out * w.to(cuda_device)).sum().backward()
    assert _rel(out.detach().cpu().numpy(), ref.detach().numpy()) < 1e-4
    assert _rel(tok.grad.cpu().numpy(), tok_ref.grad.numpy()) < 1e-3
    for k, p in enc.named_parameters():
        assert _rel(p.grad.cpu().numpy(), sd[k].grad.numpy()) < 1e-3, k


def test_transformer_and_pose_head_full_size_match_library_ops(cuda_device):
    """BASELINE.json configs[1] shapes (B=32, P=20, D=256, 8 heads, FF=1024, 4 layers; head F=256): the HIP
    kernels against PyTorch-ROCm's nn.TransformerEncoder / Linear stack on the same device, dropout off."""
    torch.manual_seed(3)
    B, P, D, H, FF, L = 32, 20, 256, 8, 1024, 4
    enc = TransformerEncoder(D, H, FF, L, norm_first=True, dropout=0.0).to(cuda_device).train()
    head = StocasticPoseRegressor(feat_dim=D, noise_dim=0).to(cuda_device).train()
    with torch.no_grad():  # non-trivial LayerNorm affine parameters and biases
        for p in list(enc.parameters()) + list(head.parameters()):
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    g = torch.Generator().manual_seed(7)
    num = torch.randint(2, P + 1, (B,), generator=g)
    valid = (torch.arange(P)[None] < num[:, None]).to(cuda_device)
    tok = (torch.randn(B, P, D, generator=g) * valid.cpu()[..., None]).to(cuda_device)
    w_r, w_t = torch.randn(B, P, 4, generator=g).to(cuda_device), torch.randn(B, P, 3, generator=g).to(cuda_device)

    def run(native):
        enc.native = head.native = native
        for p in list(enc.parameters()) + list(head.parameters()):
            p.grad = None
        x = tok.clone().requires_grad_()
        feats = enc(x, valid)
        rot, trans = head(feats)
        vm = valid[..., None].float()
        ((rot * w_r * vm).sum() + (trans * w_t * vm).sum() + 0.01 * (feats * vm).sum()).backward()
        grads = {k: p.grad.clone() for k, p in list(enc.named_parameters()) + list(head.named_parameters())}
        return feats.detach(), rot.detach(), trans.detach(), x.grad.clone(), grads

    f1, r1, t1, gx1, g1 = run(True)
    f0, r0, t0, gx0, g0 = run(False)
    v = valid.cpu().numpy()
    for a, b in ((f1, f0), (r1, r0), (t1, t0)):
        assert _rel(a.cpu().numpy()[v], b.cpu().numpy()[v]) < 1e-4
    assert _rel(gx1.cpu().numpy(), gx0.cpu().numpy()) < 1e-3
    for k in g0:
        assert _rel(g1[k].cpu().numpy(), g0[k].cpu().numpy()) < 1e-3, k
    # bit-reproducible: the same call twice gives identical gradients (fixed-order reductions)
    f2, _, _, gx2, g2 = run(True)
    assert torch.equal(f1, f2) and torch.equal(gx1, gx2) and all(torch.equal(g1[k], g2[k]) for k in g1)


def test_transformer_cross_block_hand_overs_are_stable_over_many_launches(cuda_device):
    """The split-K GEMMs hand half tiles from block to block (and XCD to XCD) with agent-scope stores ordered by a ticket —
    no fence (tf_gemm.h); the fused attention kernel hands tiles from wave to wave through LDS.  A stale tile would be a
    silent numeric error that differs from launch to launch: 300 forward + backward launches at the benchmark's shape with
    dropout ON and a fixed seed, every output and gradient bit-equal to the first launch's."""
    torch.manual_seed(11)
    B, P, D, H, FF, L = 32, 20, 256, 8, 1024, 4
    enc = TransformerEncoder(D, H, FF, L, norm_first=True, dropout=0.1).to(cuda_device).train()
    g = torch.Generator().manual_seed(5)
    num = torch.randint(2, P + 1, (B,), generator=g)
    valid = (torch.arange(P)[None] < num[:, None]).to(cuda_device)
    tok = (torch.randn(B, P, D, generator=g) * valid.cpu()[..., None]).to(cuda_device)
    w = torch.randn(B, P, D, generator=g).to(cuda_device)

    def run():
        for p in enc.parameters():
            p.grad = None
        enc._calls = 0  # the call's dropout seed is a hash of torch's seed and this counter: the same masks every time
        x = tok.clone().requires_grad_()
        feats = enc(x, valid)
        (feats * w * valid[..., None].float()).sum().backward()
        return [feats.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in enc.parameters()]

    first = run()
    assert all(torch.isfinite(t).all() for t in first)
    for it in range(300):
        again = run()
        for k, (a, b) in enumerate(zip(first, again)):
            assert torch.equal(a, b), (it, k)


@pytest.mark.parametrize("B,P", [(1, 1), (3, 2), (5, 31), (2, 32), (33, 7), (40, 20)])
def test_transformer_fused_kernels_at_the_edges_of_their_envelope(cuda_device, B, P):
    """attn_qkv_fwd_kernel / attn_do_bwd_kernel hold a sample's tokens in one 32-row tile and hand tiles between blocks per
    sample: one token, a full tile, more samples than row tiles, a batch that is not a multiple of anything — against
    PyTorch-ROCm's own TransformerEncoder on the same device (dropout off), every gradient."""
    torch.manual_seed(100 * B + P)
    D, H, FF, L = 256, 8, 1024, 2
    enc = TransformerEncoder(D, H, FF, L, norm_first=True, dropout=0.0).to(cuda_device).train()
    g = torch.Generator().manual_seed(B + 7 * P)
    num = torch.randint(1, P + 1, (B,), generator=g)
    valid = (torch.arange(P)[None] < num[:, None]).to(cuda_device)
    tok = (torch.randn(B, P, D, generator=g) * valid.cpu()[..., None]).to(cuda_device)
    w = torch.randn(B, P, D, generator=g).to(cuda_device)

    def run(native):
        enc.native = native
        for p_ in enc.parameters():
            p_.grad = None
        x = tok.clone().requires_grad_()
        out = enc(x, valid)
        (out * w * valid[..., None].float()).sum().backward()
        return out.detach(), x.grad.clone(), {k: p_.grad.clone() for k, p_ in enc.named_parameters()}

    o1, gx1, g1 = run(True)
    o0, gx0, g0 = run(False)
    v = valid.cpu().numpy()
    assert _rel(o1.cpu().numpy()[v], o0.cpu().numpy()[v]) < 1e-4
    assert _rel(gx1.cpu().numpy()[v], gx0.cpu().numpy()[v]) < 1e-3
    for k in g0:
        assert _rel(g1[k].cpu().numpy(), g0[k].cpu().numpy()) < 1e-3, k


_TF_KNOB_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[2])
from multi_part_assembly_amd.transformer import TransformerEncoder
dev = torch.device("cuda", 0)
torch.manual_seed(21)
B, P, D, H, FF, L = 6, 20, 256, 8, 1024, 2
enc = TransformerEncoder(D, H, FF, L, norm_first=True, dropout=0.1).to(dev).train()
g = torch.Generator().manual_seed(4)
num = torch.randint(2, P + 1, (B,), generator=g)
valid = (torch.arange(P)[None] < num[:, None]).to(dev)
tok = (torch.randn(B, P, D, generator=g) * valid.cpu()[..., None]).to(dev).requires_grad_()
w = torch.randn(B, P, D, generator=g).to(dev)
enc._calls = 0
out = enc(tok, valid)
(out * w * valid[..., None].float()).sum().backward()
arrs = {"out": out.detach().cpu().numpy(), "gtok": tok.grad.cpu().numpy()}
for k, p in enc.named_parameters():
    arrs["g_" + k] = p.grad.cpu().numpy()
np.savez(sys.argv[1], **arrs)
"""


def test_transformer_fused_and_separate_launch_paths_agree(cuda_device, tmp_path):
    """The fused kernels of round 6 (LayerNorm + qkv + attention forward; d o + LN2 backward + attention backward, with the
    d(LN1 output) product and LN1's backward chained behind it) each keep the launches they replaced behind an environment
    knob read once per process: every combination — in a process of its own — gives the default's outputs and gradients up
    to the order of the K sums (same dropout masks: the seed is fixed)."""
    import os
    import subprocess
    import sys
    root = str(Path(__file__).resolve().parent.parent)
    runs = {}
    for name, env in (("default", {}), ("no_qkvattn", {"MPA_TF_QKVATTN": "0"}), ("no_chain", {"MPA_TF_CHAIN": "0"}),
                      ("no_doattn", {"MPA_TF_DOATTN": "0"}),
                      ("all_off", {"MPA_TF_QKVATTN": "0", "MPA_TF_DOATTN": "0", "MPA_TF_CHAIN": "0", "MPA_TF_LNB": "0"})):
        out = tmp_path / f"{name}.npz"
        subprocess.run([sys.executable, "-c", _TF_KNOB_SCRIPT, str(out), root], check=True, env={**os.environ, **env},
                       timeout=300)
        runs[name] = dict(np.load(out))
    ref = runs.pop("default")
    for name, r in runs.items():
        for k in ref:
            assert _rel(r[k], ref[k]) < 2e-5, (name, k)


@pytest.mark.parametrize("arch,N", [("pointnet", 333), ("dgcnn", 200)])
def test_encoders_without_any_valid_part_give_zeros(cuda_device, arch, N):
    """A call in which every part is padding (found by tools/fuzz_parity.py: PointNet's BatchNorm divided by the zero
    count and returned NaN gradients): zero features, zero — finite — parameter gradients, running statistics untouched;
    the reference's BatchNorm would refuse the empty batch (modules/encoder/pointnet.py:45-55 on a [0, 3, N] input)."""
    from multi_part_assembly_amd.encoder import build_encoder
    torch.manual_seed(0)
    enc = build_encoder(arch, 128).to(cuda_device).train()
    before = {k: v.clone() for k, v in enc.state_dict().items() if "running" in k}
    pts = torch.randn(3, N, 3, device=cuda_device)
    out = enc.forward_parts(pts, torch.zeros(3, device=cuda_device))
    (out + 1.0).square().sum().backward()
    assert float(out.abs().max()) == 0.0
    for k, p in enc.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().max()) == 0.0, k
    for k, v in enc.state_dict().items():
        if "running" in k:
            assert torch.equal(v, before[k]), k
    # and a mixed call still matches the compacted one (the guard does not touch the regular path)
    v = torch.tensor([0.0, 1.0, 0.0], device=cuda_device)
    enc.zero_grad()
    mixed = enc.forward_parts(pts, v)
    assert float(mixed[0].abs().max()) == 0.0 and float(mixed[1].abs().max()) > 0.0


def test_pose_head_odd_input_width_matches_library_ops(cuda_device):
    """Input widths that are not multiples of 64 (semantic models append P labels and 32 noise channels; the
    refinement model appends the 7-d pose) run on the HIP head, which zero-pads its panels inside the workspace."""
    torch.manual_seed(5)
    for width, noise in ((180, 0), (128 + 20, 32), (263, 0)):
        head = StocasticPoseRegressor(feat_dim=width, noise_dim=noise).to(cuda_device).train()
        x = torch.randn(7, 5, width, device=cuda_device)
        w_r, w_t = torch.randn(7, 5, 4, device=cuda_device), torch.randn(7, 5, 3, device=cuda_device)

        def run(native):
            head.native = native
            for p in head.parameters():
                p.grad = None
            torch.manual_seed(9)  # same noise channels on both paths
            xi = x.clone().requires_grad_()
            rot, trans = head(xi)
            ((rot * w_r).sum() + (trans * w_t).sum()).backward()
            return rot.detach(), trans.detach(), xi.grad.clone(), {k: p.grad.clone() for k, p in head.named_parameters()}

        r1, t1, gx1, g1 = run(True)
        r0, t0, gx0, g0 = run(False)
        assert (width + noise) % 64 != 0
        assert _rel(r1.cpu().numpy(), r0.cpu().numpy()) < 1e-4 and _rel(t1.cpu().numpy(), t0.cpu().numpy()) < 1e-4
        assert gx1.shape == x.shape and _rel(gx1.cpu().numpy(), gx0.cpu().numpy()) < 1e-3
        for k in g0:
            assert g1[k].shape == g0[k].shape and _rel(g1[k].cpu().numpy(), g0[k].cpu().numpy()) < 1e-3, k


def test_grad_sink_direct_writes_equal_autograd_accumulation(cuda_device):
    """GradSink: backward kernels writing straight into the flat gradient buffer give bit-identical gradients
    to autograd's AccumulateGrad path, and a parameter used twice in one step falls back to accumulation."""
    from multi_part_assembly_amd.gradsink import GradSink
    from multi_part_assembly_amd.optim import FlatBuffers
    torch.manual_seed(11)
    head = StocasticPoseRegressor(feat_dim=64, noise_dim=0).to(cuda_device).train()
    enc = TransformerEncoder(64, 4, 128, 2, norm_first=True, dropout=0.0).to(cuda_device).train()
    flat = FlatBuffers(list(enc.parameters()) + list(head.parameters()))
    tok = torch.randn(3, 6, 64, device=cuda_device)
    valid = torch.ones(3, 6, dtype=torch.bool, device=cuda_device)

    def run(sink, twice):
        flat.zero_grad()
        with (sink if sink is not None else torch.enable_grad()):
            feats = enc(tok, valid)
            rot, trans = head(feats)
            loss = (rot * rot.detach().roll(1, -1)).sum() + trans.square().sum()
            if twice:
                rot2, trans2 = head(feats * 0.5)
                loss = loss + rot2.sum() + trans2.abs().sum()
            loss.backward()
        return flat.flat_grad.clone()

    fired = []
    sink = GradSink(on_ready=fired.append)
    for twice in (False, True):
        fired.clear()
        ref = run(None, twice)
        got = run(sink, twice)
        assert torch.equal(ref, got)
        n_direct = len(list(enc.parameters())) + (0 if twice else len(list(head.parameters())))
        assert len(fired) == n_direct  # the reused head fell back to AccumulateGrad
    assert GradSink.active is None


def test_transformer_takes_the_float_validity_matrix(cuda_device):
    """The models hand `part_valids` (float) to the encoder as it is: a part is real iff its entry == 1 — the reference's
    `part_valids == 1` (pn_transformer/network.py:88-92) applied inside the attention kernels — so any other value,
    not only 0, masks the part; bit-equal to the same call with the boolean mask."""
    torch.manual_seed(5)
    for P, d, heads in ((6, 64, 4), (20, 256, 8), (40, 128, 4)):  # matrix-core attention (P <= 32) and the scalar kernel
        enc = TransformerEncoder(d, heads, 2 * d, 2, norm_first=True, dropout=0.0).to(cuda_device).train()
        tok = torch.randn(3, P, d, device=cuda_device)
        valids = torch.ones(3, P, device=cuda_device)
        valids[0, P - 2:] = 0.0
        valids[1, 1] = 0.5   # not a real part under `== 1`
        valids[2, 0] = 2.0   # neither
        a = enc(tok, valids)
        b = enc(tok, valids == 1)
        assert torch.equal(a, b)
        changed = enc(tok, torch.ones(3, P, device=cuda_device))
        assert not torch.equal(a[1], changed[1]) and not torch.equal(a[2], changed[2])


def _small_cfg(z):
    d, heads, ffn, layers = (int(v) for v in z["cfg"])
    cfg = config.pn_transformer_everyday()
    cfg.model.pc_feat_dim, cfg.model.transformer_heads = d, heads
    cfg.model.transformer_feat_dim, cfg.model.transformer_layers = ffn, layers
    cfg.data.max_num_part = 5
    return cfg


@pytest.mark.parametrize("fused", [True, False])
def test_pn_transformer_step_matches_reference(golden, cuda_device, fused):
    z = golden("pn_transformer_step")
    model = build_model(_small_cfg(z))
    model.fused_loss = fused  # fused assembly-loss kernels vs per-function composition
    _load(model, z, "sd0.")
    _no_dropout(model)
    model.to(cuda_device).train()
    batch = {k[5:]: T(v).to(cuda_device) for k, v in z.items() if k.startswith("data.")}
    losses = model.forward_pass(batch, mode="train")
    losses["loss"].backward()
    for k in ("trans_loss", "rot_pt_cd_loss", "transform_pt_cd_loss", "rot_loss", "rot_pt_l2_loss", "loss"):
        np.testing.assert_allclose(float(losses[k]), float(z["loss." + k]), rtol=1e-4, err_msg=k)
    assert "part_quat" in batch  # forward_pass must not mutate the caller's dict
    for k, p in model.named_parameters():
        assert _rel(p.grad.cpu().numpy(), z["grad." + k]) < 2e-3, k
    for k, v in model.state_dict().items():
        if "running" in k:
            np.testing.assert_allclose(v.cpu().numpy(), z["sd1." + k], rtol=1e-4, atol=1e-5, err_msg=k)
    with torch.no_grad():
        feats = model._extract_part_feats(batch["part_pcs"], batch["part_valids"])
    assert _rel(feats.cpu().numpy(), z["act.pc_feats"]) < 1e-4


def test_fused_adam_matches_torch_adam(cuda_device):
    torch.manual_seed(0)
    ref_p = [torch.randn(37, 5, device=cuda_device), torch.randn(11, device=cuda_device), torch.randn(3, 3, 3, device=cuda_device)]
    for wd in (0.0, 0.01):
        mine = [torch.nn.Parameter(p.clone()) for p in ref_p]
        theirs = [torch.nn.Parameter(p.clone()) for p in ref_p]
        opt = FusedAdam(mine, lr=1e-2, weight_decay=wd)
        topt = (torch.optim.AdamW if wd > 0 else torch.optim.Adam)(theirs, lr=1e-2, weight_decay=wd)
        for step in range(5):
            opt.zero_grad()
            topt.zero_grad()
            gs = [torch.randn_like(p) for p in ref_p]
            for a, b, g in zip(mine, theirs, gs):
                a.grad.copy_(g)
                b.grad = g.clone()
            opt.step()
            topt.step()
        for a, b in zip(mine, theirs):
            np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)


def test_fused_adamw_param_groups_and_clipping_match_torch(cuda_device):
    """weight_decay > 0: AdamW with biases and normalisation weights exempt (utils/utils.py:90-125, base_model.py:
    394-404); clip_grad: Lightning's gradient_clip_val = torch.nn.utils.clip_grad_norm_ (scripts/train.py:90)."""
    from multi_part_assembly_amd.optim import FlatBuffers, decay_mask_for
    torch.manual_seed(3)

    def net():  # every parameter has a non-degenerate gradient (no shift fed into a following normalisation)
        torch.manual_seed(4)
        return torch.nn.Sequential(torch.nn.BatchNorm1d(7), torch.nn.Linear(7, 9), torch.nn.LayerNorm(9),
                                   torch.nn.Linear(9, 5)).to(cuda_device)

    mine, theirs = net(), net()
    flat = FlatBuffers(list(mine.parameters()))
    mask = decay_mask_for(mine, flat)
    assert 0 < float(mask.sum()) < flat.numel
    opt = FusedAdam(flat, lr=1e-2, weight_decay=0.1, decay_mask=mask, clip_grad=0.05)
    no_decay = [theirs[0].weight, theirs[0].bias, theirs[1].bias, theirs[2].weight, theirs[2].bias, theirs[3].bias]
    decay = [theirs[1].weight, theirs[3].weight]
    topt = torch.optim.AdamW([{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": 0.1}], lr=1e-2)
    for step in range(4):
        x = torch.randn(16, 7, device=cuda_device)
        opt.zero_grad()
        topt.zero_grad()
        k = 30.0 if step % 2 else 1e-3  # clipped and unclipped steps
        (mine(x).square().sum() * k).backward()
        (theirs(x).square().sum() * k).backward()
        torch.nn.utils.clip_grad_norm_(theirs.parameters(), 0.05)
        opt.step()  # no host sync between the steps: the step count lives on the device
        topt.step()
    for a, b in zip(mine.parameters(), theirs.parameters()):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=1e-4, atol=1e-5)


def test_trainer_step_matches_oracle_step(golden, cuda_device):
    """Full optimiser step: loss and gradients equal the oracle's (CPU autograd), and the parameter
    update is exactly Adam's first step on those gradients."""
    from oracle import nets as on

    z = golden("pn_transformer_step")
    cfg = _small_cfg(z)
    cfg.optimizer.lr_scheduler = ""  # constant lr = 1e-3
    model = build_model(cfg)
    _load(model, z, "sd0.")
    _no_dropout(model)
    model.to(cuda_device)
    trainer = Trainer(model, cfg)
    before = trainer.flat.flat_param.clone()
    batch = {k[5:]: T(v).to(cuda_device) for k, v in z.items() if k.startswith("data.")}
    loss = trainer.train_step(batch)
    np.testing.assert_allclose(float(loss), float(z["loss.loss"]), rtol=1e-4)

    sd = {k[4:]: T(v.copy()) for k, v in z.items() if k.startswith("sd0.")}
    params = {k: sd[k].requires_grad_() for k, _ in model.named_parameters()}
    cpu_batch = {k[5:]: T(v) for k, v in z.items() if k.startswith("data.")}
    losses, _ = on.pn_transformer_loss(sd, cpu_batch, cfg.model.transformer_layers, cfg.model.transformer_heads)
    losses["loss"].backward()
    for k, p in model.named_parameters():
        assert _rel(p.grad.cpu().numpy(), params[k].grad.numpy()) < 2e-3, k
    # first Adam step: m_hat = g, v_hat = g^2  ->  p -= lr * g / (|g| + eps)
    g = trainer.flat.flat_grad
    want = before - cfg.optimizer.lr * g / (g.abs() + 1e-8)
    np.testing.assert_allclose(trainer.flat.flat_param.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-7)


def _fresh_trainer(golden, cuda_device, **kw):
    z = golden("pn_transformer_step")
    cfg = _small_cfg(z)
    cfg.optimizer.lr_scheduler = ""
    model = build_model(cfg)
    _load(model, z, "sd0.")
    _no_dropout(model)
    model.to(cuda_device)
    batch = {k[5:]: T(v).to(cuda_device) for k, v in z.items() if k.startswith("data.")}
    return Trainer(model, cfg, **kw), batch


def _same_trajectory(a, b):
    """The pipeline is deterministic (no atomics on this path) and the graph-mode Adam uses the same
    arithmetic as the eager one, so the two trajectories must agree to the last bit."""
    assert torch.equal(a, b), float((a - b).abs().max())


def test_graph_replay_equals_eager_steps(golden, cuda_device):
    """The captured HIP graph of the whole step (zero_grad + forward + backward + Adam with device-side
    scalars) must walk the same parameter trajectory as eager launches."""
    eager, batch = _fresh_trainer(golden, cuda_device)
    graph, _ = _fresh_trainer(golden, cuda_device, use_graph=True, graph_warmup=1)
    for step in range(5):
        le = eager.train_step(batch)
        lg = graph.train_step(batch)
        assert float(lg) == float(le)
    assert graph._graph is not None  # steps 2.. were replays
    _same_trajectory(graph.flat.flat_param, eager.flat.flat_param)


def test_graph_replays_without_host_sync(golden, cuda_device):
    """Replays queued back to back with NO host synchronisation between them (the step is GPU-bound, so the host
    runs steps ahead): Adam's step count and bias corrections are advanced on the device, so each queued update
    sees its own step — the trajectory equals eager steps, bit for bit."""
    eager, batch = _fresh_trainer(golden, cuda_device)
    graph, _ = _fresh_trainer(golden, cuda_device, use_graph=True, graph_warmup=1)
    for _ in range(2):  # settle + capture
        graph.train_step(batch)
    torch.cuda.synchronize()
    for _ in range(8):
        graph.train_step(batch)  # nothing reads a result back inside this loop
    torch.cuda.synchronize()
    for _ in range(10):
        eager.train_step(batch)
    _same_trajectory(graph.flat.flat_param, eager.flat.flat_param)


def test_graph_trainer_data_parallel_path(golden, cuda_device):
    """world > 1 code path of graph mode on one GPU: a 1-rank RCCL group with the trainer told it is one of
    two ranks — the graph then holds forward+backward only, the all-reduce and Adam (grad_scale = 1/2) run
    behind it.  Equals an eager single-rank trainer whose optimiser scales gradients by 1/2."""
    import os
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=cuda_device)
    try:
        dp, batch = _fresh_trainer(golden, cuda_device, use_graph=True, graph_warmup=1)
        dp.world = 2
        ref, _ = _fresh_trainer(golden, cuda_device)
        for step in range(4):
            dp.train_step(batch)
            ref.model.train()
            ref.optimizer.zero_grad()
            ref.model.training_step(batch).backward()
            ref.optimizer.grad_scale = 0.5
            ref.optimizer.step()
        _same_trajectory(dp.flat.flat_param, ref.flat.flat_param)
    finally:
        dist.destroy_process_group()


# ---- regressions for the round-2 advisor findings ---------------------------------------------------------------------------
def test_second_backward_over_a_consumed_workspace_raises(cuda_device):
    """The DGCNN / MLP-layer / assembly-loss backward passes overwrite their saved workspaces; a second backward over the
    same forward must raise instead of returning gradients computed from clobbered buffers."""
    from multi_part_assembly_amd.mlp import mlp_layer
    enc = build_encoder("dgcnn", 64).to(cuda_device).train()
    x = (torch.randn(2, 64, 3, device=cuda_device) * 0.2).requires_grad_()
    out = enc(x)
    out.sum().backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="second backward"):
        out.sum().backward()
    lin = torch.nn.Linear(64, 64).to(cuda_device)
    y = mlp_layer(torch.randn(128, 64, device=cuda_device, requires_grad=True), lin.weight, lin.bias, None, relu=True)
    y.sum().backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="second backward"):
        y.sum().backward()


def test_optimizer_clip_and_step_count_can_be_assigned_late(cuda_device):
    """`clip_grad` set after the first step allocates its workspace on first use; assigning `step_count` re-synchronises
    the device-side counter (bias corrections follow the host mirror)."""
    torch.manual_seed(0)
    p = torch.nn.Parameter(torch.randn(1000, device=cuda_device))
    q = torch.nn.Parameter(p.detach().clone())
    mine, ref = FusedAdam([p], lr=1e-2), torch.optim.Adam([q], lr=1e-2)
    for step in range(3):
        g = torch.randn(1000, device=cuda_device) * 10
        p.grad.copy_(g) if p.grad is not None else setattr(p, "grad", g.clone())
        q.grad = g.clone()
        if step == 1:
            mine.clip_grad = 1.0  # assigned after the first step
        if step >= 1:
            torch.nn.utils.clip_grad_norm_([q], 1.0)
        mine.step()
        ref.step()
    np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=2e-5, atol=2e-6)
    mine.step_count = 10  # e.g. a resumed run
    ref.state[q]["step"].fill_(10)
    g = torch.randn(1000, device=cuda_device)
    p.grad.copy_(g)
    q.grad = g.clone()
    mine.clip_grad = None
    mine.step()
    ref.step()
    assert mine.step_count == 11
    np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=2e-5, atol=2e-6)


def test_gru_reports_grid_residency(cuda_device):
    """csrc/gru.hip's step barrier needs its whole grid resident: `supported()` asks the runtime's occupancy calculator
    (mpa_gru_resident) and is true for the shipped shapes on a whole MI355X, false for shapes the kernels are not built for."""
    from multi_part_assembly_amd.gru import supported
    assert supported(256, 32) and supported(128, 3)
    assert not supported(192, 32) and not supported(256, 65)


def test_two_trainers_walk_the_same_trajectory_at_the_benchmark_size(cuda_device):
    """Run-to-run reproducibility of the whole step at BASELINE.json configs[1]'s size (B = 32, P = 20, N = 1000): two
    trainers with identical initial weights take 600 steps side by side over four rotated batches; their parameter
    buffers must stay bit-equal after every step.  (tools/exp_race_hunt.py is the long form — it caught an LDS race in a
    block's FIRST unit of the layer-2 backward kernel, one diverging step in ~300, that no single-step parity test and no
    run-twice check at small sizes had seen: only a block's first unit at full occupancy was exposed.)"""
    from multi_part_assembly_amd import synthetic
    cfg = config.pn_transformer_everyday()
    batches = []
    for k in range(4):
        bt = synthetic.make_batch(32, 20, 1000, preset="everyday", seed=1234 + 1000 * k, device=cuda_device)
        bt.pop("num_parts")
        batches.append(bt)

    def make():
        torch.manual_seed(0)
        return Trainer(build_model(cfg).to(cuda_device), cfg)

    a, b = make(), make()
    assert torch.equal(a.flat.flat_param, b.flat.flat_param)
    for i in range(600):
        a.train_step(batches[i % 4], i)
        b.train_step(batches[i % 4], i)
        if i % 20 == 19 or i < 20:
            assert torch.equal(a.flat.flat_param, b.flat.flat_param), f"trajectories diverge by step {i}"
    assert torch.equal(a.flat.flat_param, b.flat.flat_param)

