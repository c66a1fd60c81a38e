"""Parity of the HIP pose-transform kernels and the loss functions built on them with the golden
fixtures captured from the reference (utils/transforms.py, utils/loss.py) and with the oracle."""
import numpy as np
import pytest
import torch

from multi_part_assembly_amd import loss as L
from multi_part_assembly_amd import transforms as TR
from multi_part_assembly_amd.rotation import Rotation3D

pytestmark = pytest.mark.gpu


def _dev(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def test_rotation3d_and_transforms_bit_exact(golden, cuda_device):
    z = golden("transforms")
    rot = Rotation3D(_dev(z["quat_in"], cuda_device))
    np.testing.assert_array_equal(rot.rot.cpu().numpy(), z["quat_checked"])
    pc, t = _dev(z["pc"], cuda_device), _dev(z["trans"], cuda_device)
    np.testing.assert_array_equal(TR.rot_pc(rot, pc).cpu().numpy(), z["rot_pc"])
    np.testing.assert_array_equal(TR.transform_pc(t, rot, pc).cpu().numpy(), z["transform_pc"])
    flat = TR.qrot(rot.rot.reshape(-1, 4), pc[:, :, 0].reshape(-1, 3))
    np.testing.assert_array_equal(flat.cpu().numpy(), z["qrot_flat"])
    # raw-tensor form of the API
    np.testing.assert_array_equal(TR.rot_pc(rot.rot, pc, rot_type="quat").cpu().numpy(), z["rot_pc"])


def test_pose_apply_gradients_match_autograd_of_definition(cuda_device):
    from oracle import geometry as og

    g = torch.Generator().manual_seed(4)
    q = torch.randn(3, 5, 4, generator=g)
    t = torch.randn(3, 5, 3, generator=g)
    pc = torch.randn(3, 5, 33, 3, generator=g)
    w = torch.randn(3, 5, 33, 3, generator=g)
    leaves = [x.clone().double().requires_grad_() for x in (q, t, pc)]
    (og.transform_pc(leaves[1], leaves[0], leaves[2]) * w.double()).sum().backward()
    dl = [x.clone().to(cuda_device).requires_grad_() for x in (q, t, pc)]
    (TR.transform_pc(dl[1], dl[0], dl[2], rot_type="quat") * w.to(cuda_device)).sum().backward()
    for a, b in zip(dl, leaves):
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=2e-5, atol=2e-5)


LOSSES = ["trans_l2", "rot_cosine", "rot_l2", "rot_points_l2", "rot_points_cd", "shape_cd_train",
          "shape_cd_eval"]


@pytest.mark.parametrize("name", LOSSES)
def test_losses_match_reference(golden, cuda_device, name):
    z = golden("losses")
    d = lambda k: _dev(z[k], cuda_device)
    pts, valids, tg = d("pts"), d("valids"), d("trans_gt")
    rg = Rotation3D(d("quat_gt"))
    qp = d("quat_pred").requires_grad_()
    tp = d("trans_pred").requires_grad_()
    rp = Rotation3D(qp)
    fn = {
        "trans_l2": lambda: L.trans_l2_loss(tp, tg, valids),
        "rot_cosine": lambda: L.rot_cosine_loss(rp, rg, valids),
        "rot_l2": lambda: L.rot_l2_loss(rp, rg, valids),
        "rot_points_l2": lambda: L.rot_points_l2_loss(pts, rp, rg, valids),
        "rot_points_cd": lambda: L.rot_points_cd_loss(pts, rp, rg, valids, ret_pts=True),
        "shape_cd_train": lambda: L.shape_cd_loss(pts, tp, tg, rp, rg, valids, ret_pts=True, training=True),
        "shape_cd_eval": lambda: L.shape_cd_loss(pts, tp, tg, rp, rg, valids, training=False),
    }[name]
    res = fn()
    extra = ()
    if isinstance(res, tuple):
        res, *extra = res
    (res * d("w")).sum().backward()
    # north_star bar: 1e-4 relative in fp32
    np.testing.assert_allclose(res.detach().cpu().numpy(), z[name], rtol=1e-5, atol=1e-7)
    gq = qp.grad.cpu().numpy() if qp.grad is not None else np.zeros_like(z["quat_pred"])
    gt = tp.grad.cpu().numpy() if tp.grad is not None else np.zeros_like(z["trans_pred"])
    np.testing.assert_allclose(gq, z[name + "_gquat"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(gt, z[name + "_gtrans"], rtol=1e-4, atol=2e-6)
    for i, e in enumerate(extra):  # the transformed clouds are bit-identical to the reference's
        np.testing.assert_array_equal(e.detach().cpu().numpy(), z[f"{name}_pts{i + 1}"])


# ---- fused assembly loss (csrc/assembly_loss.hip) ---------------------------------------------------
FUSED = {"trans_loss": "trans_l2", "rot_loss": "rot_cosine", "rot_pt_l2_loss": "rot_points_l2",
         "rot_pt_cd_loss": "rot_points_cd", "transform_pt_cd_loss": "shape_cd_train"}


@pytest.mark.parametrize("term", list(FUSED) + ["transform_pt_cd_loss/eval"])
def test_fused_loss_matches_reference(golden, cuda_device, term):
    z = golden("losses")
    d = lambda k: _dev(z[k], cuda_device)
    training = not term.endswith("/eval")
    name = term.split("/")[0]
    ref = FUSED[name] if training else "shape_cd_eval"
    qp = d("quat_pred").requires_grad_()
    tp = d("trans_pred").requires_grad_()
    terms, pts = L.geometric_assembly_loss(d("pts"), tp, Rotation3D(qp), d("trans_gt"),
                                           Rotation3D(d("quat_gt")), d("valids"), training=training,
                                           ret_pts=True)
    (terms[name] * d("w")).sum().backward()
    np.testing.assert_allclose(terms[name].detach().cpu().numpy(), z[ref], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(qp.grad.cpu().numpy(), z[ref + "_gquat"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(tp.grad.cpu().numpy(), z[ref + "_gtrans"], rtol=1e-4, atol=2e-6)
    if name == "transform_pt_cd_loss" and training:  # ret_pts clouds, padded parts fully filled
        np.testing.assert_array_equal(pts[0].cpu().numpy(), z["shape_cd_train_pts1"])
        np.testing.assert_array_equal(pts[1].cpu().numpy(), z["shape_cd_train_pts2"])


def test_fused_loss_equals_composed_path_at_full_size(cuda_device):
    """B=8, P=20, N=1000 with padded parts: the fused kernels (pad representatives, skipped padded
    queries) must give the same five terms and the same pose gradients as composing the generic
    pose / Chamfer operators over all P*N slots."""
    from multi_part_assembly_amd import synthetic

    batch = synthetic.make_batch(8, 20, 1000, seed=5, device=cuda_device, num_parts=[2, 20, 7, 11, 3, 15, 9, 19])
    g = torch.Generator().manual_seed(1)
    qp = torch.nn.functional.normalize(torch.randn(8, 20, 4, generator=g), dim=-1).to(cuda_device)
    tp = (torch.randn(8, 20, 3, generator=g) * 0.2).to(cuda_device)
    w = (torch.rand(5, 8, generator=g) + 0.5).to(cuda_device)
    pcs, v = batch["part_pcs"], batch["part_valids"]
    rg, tg = Rotation3D(batch["part_quat"]), batch["part_trans"]

    def run(fused):
        q = qp.clone().requires_grad_()
        t = tp.clone().requires_grad_()
        rp = Rotation3D(q)
        if fused:
            terms, _ = L.geometric_assembly_loss(pcs, t, rp, tg, rg, v, training=True)
            vals = [terms[k] for k in L.LOSS_TERMS]
        else:
            vals = [L.trans_l2_loss(t, tg, v), L.rot_points_cd_loss(pcs, rp, rg, v),
                    L.shape_cd_loss(pcs, t, tg, rp, rg, v, training=True),
                    L.rot_cosine_loss(rp, rg, v), L.rot_points_l2_loss(pcs, rp, rg, v)]
        sum((x * w[i]).sum() for i, x in enumerate(vals)).backward()
        return [x.detach().cpu().numpy() for x in vals], q.grad.cpu().numpy(), t.grad.cpu().numpy()

    lf, gqf, gtf = run(True)
    lc, gqc, gtc = run(False)
    for a, b, name in zip(lf, lc, L.LOSS_TERMS):
        np.testing.assert_allclose(a, b, rtol=2e-6, atol=1e-8, err_msg=name)
    np.testing.assert_allclose(gqf, gqc, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(gtf, gtc, rtol=1e-4, atol=1e-6)


def _raw_assembly_forward(batch, qp, tp, mode, monkeypatch, with_part=False, part="gate"):
    """Call the C ABI directly so that the arg-min arrays in the int workspace can be inspected.  `part`: the per-part
    Chamfer on the matrix-core gated search (csrc/gate_nn.hip, the default) or on the scan / leaf search of `mode`."""
    import ctypes
    from multi_part_assembly_amd import _lib

    monkeypatch.setenv("MPA_SHAPE_SEARCH", mode)
    monkeypatch.setenv("MPA_PART_SEARCH", part)
    pcs, v = batch["part_pcs"], batch["part_valids"]
    qg, tg = Rotation3D(batch["part_quat"]).rot.contiguous(), batch["part_trans"].contiguous()
    B, P, N, _ = pcs.shape
    L_ = _lib.lib()
    nf, ni = ctypes.c_int64(), ctypes.c_int64()
    _lib.check(L_.mpa_assembly_loss_workspace(B, P, N, ctypes.byref(nf), ctypes.byref(ni)), "ws")
    # poison the workspaces: nothing may depend on what a previous user left in them
    fws = torch.full((nf.value,), float("nan"), device=pcs.device)
    iws = torch.full((ni.value,), 0x7F7F7F7F, dtype=torch.int32, device=pcs.device)
    losses = torch.empty(5, B, device=pcs.device)
    st = L_.mpa_assembly_loss_forward(_lib.ptr(pcs), _lib.ptr(v), _lib.ptr(qp), _lib.ptr(tp), _lib.ptr(qg),
                                      _lib.ptr(tg), B, P, N, 1, 0, _lib.ptr(fws), _lib.ptr(iws),
                                      _lib.ptr(losses), _lib.current_stream(pcs.device))
    _lib.check(st, "fwd")
    torch.cuda.synchronize()
    pn = B * P * N
    if with_part:
        return losses, [iws[k * pn:(k + 1) * pn].view(B, P, N) for k in range(4)]
    return losses, iws[2 * pn:3 * pn].view(B, P, N), iws[3 * pn:4 * pn].view(B, P, N)


def _assert_searches_agree(batch, qp, tp, monkeypatch):
    """brute-force scan == grid-pruned search == leaf search == per-sample choice of the two: all four arg-min arrays (per-part Chamfer both ways,
    whole-shape Chamfer both ways) bit-equal on the valid parts, the five loss terms to summation order."""
    lb, ib = _raw_assembly_forward(batch, qp, tp, "brute", monkeypatch, with_part=True, part="scan")
    valid = batch["part_valids"].bool()
    # (auto: per sample, the grid or the leaves; part = gate: the per-part term on the matrix-core gated search, scan: on
    # the exhaustive scan (brute, grid) / the leaf search (leaf, auto))
    for mode, part in (("brute", "gate"), ("grid", "gate"), ("grid", "scan"), ("leaf", "gate"), ("leaf", "scan"),
                       ("auto", "gate"), ("auto", "scan")):
        lm, im = _raw_assembly_forward(batch, qp, tp, mode, monkeypatch, with_part=True, part=part)
        for k in range(4):
            assert torch.equal(ib[k][valid], im[k][valid]), (mode, part, k, int((ib[k][valid] != im[k][valid]).sum()))
        np.testing.assert_allclose(lm.cpu().numpy(), lb.cpu().numpy(), rtol=2e-6, atol=1e-9, err_msg=f"{mode}/{part}")
    monkeypatch.delenv("MPA_PART_SEARCH", raising=False)


def test_fused_loss_at_benchmark_size_matches_oracle(cuda_device, monkeypatch):
    """The benchmark batch itself (B = 32, P = 20, N = 1000, seed 1234 — bench.py's rank-0 shard) against the CPU
    oracle: all five loss terms and the pose gradients within 1e-4 relative, and the arg-min arrays of both Chamfer
    searches BIT-EQUAL to the C oracle's exhaustive scan over the same transformed clouds (padded parts filled with
    1e3 before the transform, utils/loss.py:173-175).  2.6e10 pair evaluations on the host: slow (seconds on the GPU
    box's cores), but it is the only HIP-vs-oracle check at the size the metric is quoted on."""
    from multi_part_assembly_amd import synthetic
    from oracle import chamfer as oc
    from oracle import geometry as og

    B, P, N = 32, 20, 1000
    batch = synthetic.make_batch(B, P, N, preset="everyday", seed=1234, device=cuda_device)
    g = torch.Generator().manual_seed(99)
    qp = torch.nn.functional.normalize(torch.randn(B, P, 4, generator=g), dim=-1)
    tp = torch.randn(B, P, 3, generator=g) * 0.3
    w = torch.rand(5, B, generator=g) + 0.5
    pcs, v = batch["part_pcs"], batch["part_valids"]
    # HIP: fused loss + backward
    q = qp.to(cuda_device).requires_grad_()
    t = tp.to(cuda_device).requires_grad_()
    terms, _ = L.geometric_assembly_loss(pcs, t, Rotation3D(q), batch["part_trans"], Rotation3D(batch["part_quat"]),
                                         v, training=True)
    sum((terms[k] * w[i].to(cuda_device)).sum() for i, k in enumerate(L.LOSS_TERMS)).backward()
    # oracle: same definitions on the CPU
    cq, ct = qp.clone().requires_grad_(), tp.clone().requires_grad_()
    cpcs, cv = pcs.cpu(), v.cpu()
    gq, gtr = og.checked_quat(batch["part_quat"].cpu()), batch["part_trans"].cpu()
    ref = og.calc_loss_geometric(og.checked_quat(cq), ct, cpcs, gq, gtr, cv, training=True)
    sum((ref[k] * w[i]).sum() for i, k in enumerate(L.LOSS_TERMS)).backward()
    for k in L.LOSS_TERMS:
        np.testing.assert_allclose(terms[k].detach().cpu().numpy(), ref[k].detach().numpy(), rtol=1e-4, atol=1e-7,
                                   err_msg=k)
    scale_q, scale_t = float(cq.grad.abs().max()), float(ct.grad.abs().max())
    assert float((q.grad.cpu() - cq.grad).abs().max()) < 1e-4 * scale_q
    assert float((t.grad.cpu() - ct.grad).abs().max()) < 1e-4 * scale_t
    # arg-min arrays of the two searches, straight from the workspace of the C ABI
    _, s1, s2 = _raw_assembly_forward(batch, qp.to(cuda_device).contiguous(), tp.to(cuda_device).contiguous(), "leaf",
                                      monkeypatch)
    for other in ("grid", "auto"):
        _, g1, g2 = _raw_assembly_forward(batch, qp.to(cuda_device).contiguous(), tp.to(cuda_device).contiguous(), other,
                                          monkeypatch)
        assert torch.equal(s1[v.bool()], g1[v.bool()]) and torch.equal(s2[v.bool()], g2[v.bool()]), other
    filled = cpcs.masked_fill(cv[..., None, None] == 0, 1e3)
    c1 = og.transform_pc(tp, og.checked_quat(qp), filled).flatten(1, 2).numpy()
    c2 = og.transform_pc(gtr, gq, filled).flatten(1, 2).numpy()
    _, i1, _, i2 = oc.chamfer_forward(c1, c2)
    valid = cv.bool()
    assert torch.equal(s1.cpu()[valid].long(), torch.from_numpy(i1).view(B, P, N)[valid])
    assert torch.equal(s2.cpu()[valid].long(), torch.from_numpy(i2).view(B, P, N)[valid])


@pytest.mark.parametrize("spread", [0.05, 0.6, 30.0])
def test_grid_pruned_search_is_bit_identical_to_brute_force(cuda_device, monkeypatch, spread):
    """The spatially pruned whole-shape search must return exactly the brute-force arg-mins (same distance
    arithmetic, lowest index on ties) — for predicted poses that pile all parts at the origin (spread 0.05),
    spread them like the ground truth (0.6) or throw them far outside the target cloud (30)."""
    from multi_part_assembly_amd import synthetic

    batch = synthetic.make_batch(8, 20, 1000, seed=11, device=cuda_device, num_parts=[1, 20, 2, 13, 5, 17, 9, 20])
    g = torch.Generator().manual_seed(int(spread * 100))
    qp = torch.nn.functional.normalize(torch.randn(8, 20, 4, generator=g), dim=-1).to(cuda_device)
    tp = (torch.randn(8, 20, 3, generator=g) * spread).to(cuda_device)
    _assert_searches_agree(batch, qp, tp, monkeypatch)


@pytest.mark.parametrize("B", [1, 3, 5, 13])
def test_grid_search_wave_plan_covers_every_batch_size(cuda_device, monkeypatch, B):
    """The search waves get their (sample, direction) from an XCD-aware plan (grid_assign_kernel: pairs dealt to the 8
    XCDs by descending work, wave shares proportional to work) when there are at least 8 pairs, and from the plain
    block mapping below that: batch sizes on both sides of the switch, pair counts that are no multiple of 8, very
    uneven samples and a sample with no valid part at all must give the brute-force arg-mins."""
    from multi_part_assembly_amd import synthetic

    counts = [20, 1, 0, 7, 20, 2, 13, 0, 5, 20, 1, 9, 3][:B]
    batch = synthetic.make_batch(B, 20, 500, seed=23 + B, device=cuda_device, num_parts=[max(c, 1) for c in counts])
    for b, c in enumerate(counts):
        if c == 0:  # no valid part in this sample
            batch["part_valids"][b] = 0
    g = torch.Generator().manual_seed(B)
    qp = torch.nn.functional.normalize(torch.randn(B, 20, 4, generator=g), dim=-1).to(cuda_device)
    tp = (torch.randn(B, 20, 3, generator=g) * 0.4).to(cuda_device)
    _assert_searches_agree(batch, qp, tp, monkeypatch)


def test_grid_pruned_search_handles_duplicates_and_flat_clouds(cuda_device, monkeypatch):
    """Degenerate geometry: every part is the same flat (z = 0) lattice patch, so there are exact distance
    ties everywhere and the grid is one cell thick — indices must still match the in-order scan."""
    B, P, N = 8, 4, 256
    xs = torch.arange(16.0).repeat_interleave(16) * 0.05
    ys = torch.arange(16.0).repeat(16) * 0.05
    part = torch.stack([xs, ys, torch.zeros(256)], -1)
    pcs = part[None, None].repeat(B, P, 1, 1).to(cuda_device).contiguous()
    v = torch.ones(B, P, device=cuda_device)
    v[:, 3] = 0
    ident = torch.tensor([1.0, 0, 0, 0], device=cuda_device).repeat(B, P, 1)
    batch = {"part_pcs": pcs, "part_valids": v, "part_quat": ident.clone(),
             "part_trans": torch.zeros(B, P, 3, device=cuda_device)}
    tp = torch.zeros(B, P, 3, device=cuda_device)
    tp[:, 1, 0] = 0.05  # part 1 shifted by exactly one lattice step
    _assert_searches_agree(batch, ident.contiguous(), tp, monkeypatch)


@pytest.mark.parametrize("N,P", [(1, 3), (16, 5), (33, 4), (100, 20), (257, 7), (1100, 6), (2048, 2), (2049, 2)])
def test_leaf_search_point_counts_and_masks(cuda_device, monkeypatch, N, P):
    """The leaf search pads every part to a power of two >= 32 slots (pad slots at the end of the last leaf, whole leaves
    of padding for N just above a power of two) and falls back to the grid beyond 2048 points: sizes on every side of
    those edges, with NON-PREFIX valid masks and a sample without any valid part."""
    from multi_part_assembly_amd import synthetic

    B = 5
    batch = synthetic.make_batch(B, P, N, seed=100 + N, device=cuda_device, num_parts=[P] * B)
    g = torch.Generator().manual_seed(N)
    mask = (torch.rand(B, P, generator=g) < 0.6).float()
    mask[0] = 1.0
    mask[1] = 0.0
    mask[2, 0] = 0.0
    mask[2, -1] = 1.0
    batch["part_valids"] = mask.to(cuda_device)
    qp = torch.nn.functional.normalize(torch.randn(B, P, 4, generator=g), dim=-1).to(cuda_device)
    tp = (torch.randn(B, P, 3, generator=g) * 0.3).to(cuda_device)
    _assert_searches_agree(batch, qp, tp, monkeypatch)


@pytest.mark.parametrize("B,P,N", [(1, 3, 33), (1, 5, 101), (3, 1, 7)])
def test_leaf_search_with_odd_total_point_count(cuda_device, monkeypatch, B, P, N):
    """B * P * N odd: the grid region of the workspace used to end 8 bytes past a 16-byte boundary, which left the leaf
    search's float4 arrays behind it misaligned (now rounded up in grid_workspace_floats) — leaf and auto against the scan."""
    from multi_part_assembly_amd import synthetic

    assert (B * P * N) % 2 == 1
    batch = synthetic.make_batch(B, P, N, seed=7 + N, device=cuda_device, num_parts=[P] * B)
    g = torch.Generator().manual_seed(N)
    qp = torch.nn.functional.normalize(torch.randn(B, P, 4, generator=g), dim=-1).to(cuda_device)
    tp = (torch.randn(B, P, 3, generator=g) * 0.3).to(cuda_device)
    _assert_searches_agree(batch, qp, tp, monkeypatch)


def test_leaf_search_with_trained_poses_and_given_order(cuda_device, monkeypatch):
    """Predictions within 2 % of the ground truth (the regime the twin-point seed is for), the order computed ONCE by
    `part_order` and handed to two evaluations with different poses: identical to the evaluation that orders itself and
    to the brute-force scan."""
    from multi_part_assembly_amd import synthetic

    batch = synthetic.make_batch(6, 20, 1000, seed=5, device=cuda_device, num_parts=[20, 3, 11, 1, 17, 8])
    pcs, v = batch["part_pcs"], batch["part_valids"]
    g = torch.Generator().manual_seed(8)
    order = L.part_order(pcs, v)
    for noise in (0.02, 0.5):
        qp = torch.nn.functional.normalize(batch["part_quat"].cpu() + noise * torch.randn(6, 20, 4, generator=g), dim=-1)
        qp = torch.where(v.cpu()[..., None] > 0, qp, torch.tensor([1.0, 0, 0, 0])).to(cuda_device).contiguous()
        tp = (batch["part_trans"].cpu() + noise * torch.randn(6, 20, 3, generator=g)).to(cuda_device).contiguous()
        _assert_searches_agree(batch, qp, tp, monkeypatch)
        monkeypatch.delenv("MPA_SHAPE_SEARCH", raising=False)
        a, _ = L.geometric_assembly_loss(pcs, tp, Rotation3D(qp), batch["part_trans"], Rotation3D(batch["part_quat"]), v,
                                         search="leaf")
        b, _ = L.geometric_assembly_loss(pcs, tp, Rotation3D(qp), batch["part_trans"], Rotation3D(batch["part_quat"]), v,
                                         order=order, search="leaf")
        g_, _ = L.geometric_assembly_loss(pcs, tp, Rotation3D(qp), batch["part_trans"], Rotation3D(batch["part_quat"]), v)
        for k in L.LOSS_TERMS:  # (the default search, the grid of rounds 1-4: same terms to summation order)
            np.testing.assert_allclose(g_[k].cpu().numpy(), a[k].cpu().numpy(), rtol=2e-6, atol=1e-9, err_msg=k)
        for k in L.LOSS_TERMS:
            assert torch.equal(a[k], b[k]), k


def test_auto_search_routes_samples_both_ways(cuda_device, monkeypatch):
    """One batch whose samples fall on both sides of the routing rule (few large parts -> grid, many small parts -> leaf
    search): the per-sample choice must give the brute-force answer for every sample."""
    from multi_part_assembly_amd import synthetic

    a = synthetic.make_batch(4, 20, 500, preset="everyday", seed=31, device=cuda_device, num_parts=[3, 20, 9, 14])
    b = synthetic.make_batch(4, 20, 500, preset="artifact", seed=32, device=cuda_device, num_parts=[20, 12, 16, 13])
    batch = {k: torch.cat([a[k], b[k]])[[0, 4, 1, 5, 2, 6, 3, 7]].contiguous()
             for k in ("part_pcs", "part_valids", "part_quat", "part_trans")}
    g = torch.Generator().manual_seed(5)
    qp = torch.nn.functional.normalize(torch.randn(8, 20, 4, generator=g), dim=-1).to(cuda_device)
    tp = (torch.randn(8, 20, 3, generator=g) * 0.2).to(cuda_device)
    _assert_searches_agree(batch, qp, tp, monkeypatch)


def test_leaf_search_survives_non_finite_poses(cuda_device, monkeypatch):
    """A diverged step hands the loss NaN / inf poses: the searches must terminate and agree with the brute-force scan
    (NaN candidates never win; a query without any candidate below 1e32 keeps index -1)."""
    from multi_part_assembly_amd import synthetic

    batch = synthetic.make_batch(4, 6, 200, seed=3, device=cuda_device, num_parts=[6, 2, 4, 5])
    g = torch.Generator().manual_seed(1)
    qp = torch.nn.functional.normalize(torch.randn(4, 6, 4, generator=g), dim=-1)
    tp = torch.randn(4, 6, 3, generator=g) * 0.3
    tp[0, 1, 0] = float("nan")
    tp[1, 0, 2] = float("inf")
    qp[2, 3, 1] = float("nan")
    tp[3, 2] = 1e20
    _assert_searches_agree(batch, qp.to(cuda_device).contiguous(), tp.to(cuda_device).contiguous(), monkeypatch)


def test_rotation3d_constructor_rule_on_device(cuda_device):
    """Quaternions of norm <= 0.5 (zero padding) become the identity, gradients pass only where the input was kept
    (rotation.py:115-126 upstream): the one-launch HIP path against the tensor-op composition on the CPU."""
    from multi_part_assembly_amd.rotation import Rotation3D
    g = torch.Generator().manual_seed(2)
    q = torch.randn(4, 7, 4, generator=g)
    q[0, 0] = 0.0
    q[1, 2] = torch.tensor([0.3, 0.0, 0.4, 0.0])    # norm exactly 0.5 -> identity
    q[2, 3] = torch.tensor([0.3, 0.0, 0.4, 1e-3])   # just above
    w = torch.randn(4, 7, 4, generator=g)
    a = q.clone().requires_grad_()
    b = q.clone().to(cuda_device).requires_grad_()
    (Rotation3D(a).rot * w).sum().backward()
    rb = Rotation3D(b).rot
    (rb * w.to(cuda_device)).sum().backward()
    assert torch.equal(rb.detach().cpu(), Rotation3D(q).rot)
    assert torch.equal(b.grad.cpu(), a.grad)
    assert torch.equal(rb[0, 0].cpu(), torch.tensor([1.0, 0, 0, 0])) and b.grad[0, 0].abs().sum() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("K,B", [(5, 32), (1, 1), (7, 200), (64, 3)])
def test_weighted_term_means_equal_mean_and_dot(K, B):
    """mpa_loss_reduce_* against the two library ops it replaces (base_model.py:348-387, one sample): the per-term batch means,
    the weighted total, and the gradient with either or both outputs used."""
    from multi_part_assembly_amd.loss import weighted_term_means
    g = torch.Generator().manual_seed(K * 1000 + B)
    terms = torch.randn(K, B, generator=g).cuda()
    w = torch.rand(K, generator=g).cuda()
    cm = torch.randn(K, generator=g).cuda()
    a = terms.clone().requires_grad_()
    means, loss = weighted_term_means(a, w)
    ref = terms.clone().requires_grad_()
    rmeans = ref.mean(dim=1)
    rloss = torch.dot(rmeans, w)
    torch.testing.assert_close(means, rmeans, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(loss, rloss, rtol=1e-5, atol=1e-6)
    (3.0 * loss + (means * cm).sum()).backward()
    (3.0 * rloss + (rmeans * cm).sum()).backward()
    torch.testing.assert_close(a.grad, ref.grad, rtol=1e-5, atol=1e-7)
    b = terms.clone().requires_grad_()
    weighted_term_means(b, w)[1].backward()  # the training path: only the total is differentiated
    torch.testing.assert_close(b.grad, (w / B)[:, None].expand(K, B), rtol=1e-6, atol=0)
    means2, loss2 = weighted_term_means(terms, w)  # fixed reduction order: bit-reproducible
    assert torch.equal(means2, means.detach()) and torch.equal(loss2, loss.detach())
