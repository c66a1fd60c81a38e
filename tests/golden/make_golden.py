#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE, in this container.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--only chamfer,transforms,...]

Needs /root/reference (read-only mount, absent on the GPU box) and the import shims of
_reference_shim.py.  Outputs are plain .npz files holding inputs and the reference's outputs —
data only, no reference source.  Every fixture is seeded; re-running reproduces the files bit for
bit on the same torch build (torch 2.10.0 CPU here).

Fixture -> reference entry points exercised
  chamfer.npz       utils/chamfer/test_chamfer.py:8-31  bpdist2 / nn_distance_torch (the reference's
                    own ground truth for ChamferForwardKernel), fp64 autograd for the backward
  transforms.npz    utils/transforms.py:75-109,199-244  qrot/qtransform/rot_pc/transform_pc,
                    utils/rotation.py:115-167           Rotation3D (zero-quaternion handling)
  losses.npz        utils/loss.py:7-202                 every loss on the geometric path + input grads
  pointnet.npz      models/modules/encoder/pointnet.py:6-41
  dgcnn.npz         models/modules/encoder/dgcnn.py:8-109
  dgcnn_graphs.npz  models/modules/encoder/dgcnn.py:8-15,84-96  the inputs of all four EdgeConv stages and the
                    reference's own `knn(x, 20)` output for each, on the dgcnn.npz cloud (3 x 96 points) and
                    on a 2 x 1000-point cloud (the benchmark's points per part)
  transformer.npz   models/pn_transformer/transformer.py:37-79, models/modules/regressor.py:30-84
  pn_transformer_step.npz  models/pn_transformer/network.py:70-139 + models/modules/base_model.py
                    forward_pass/loss_function/_calc_loss on a seeded synthetic batch (+ all grads)
"""
from __future__ import annotations

import argparse
import importlib
import os
import sys
from pathlib import Path

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import _reference_shim as shim  # noqa: E402

torch.set_num_threads(4)
torch.use_deterministic_algorithms(True)


def npy(t):
    if hasattr(t, "rot"):  # Rotation3D
        t = t.rot
    return t.detach().cpu().numpy().copy()  # copy: buffers are updated in place later on


def save(name, **arrays):
    path = HERE / f"{name}.npz"
    np.savez_compressed(path, **arrays)
    print(f"wrote {path.name}: {len(arrays)} arrays, {path.stat().st_size / 1024:.1f} KiB")


def unit_quats(g, *shape):
    q = torch.randn(*shape, 4, generator=g)
    return F.normalize(q, dim=-1)


# --------------------------------------------------------------------------------------------------
def gen_chamfer():
    _, nn_distance_torch = shim.load_reference_bruteforce()
    g = torch.Generator().manual_seed(1001)
    out = {}
    cases = {"a": (2, 64, 64), "b": (3, 100, 77), "c": (1, 1000, 1000), "d": (1, 1100, 900)}
    for k, (B, n1, n2) in cases.items():
        x1 = torch.rand(B, n1, 3, generator=g)
        x2 = torch.rand(B, n2, 3, generator=g)
        d1, i1, d2, i2 = nn_distance_torch(x1, x2, "NWC")
        out.update({f"{k}_xyz1": npy(x1), f"{k}_xyz2": npy(x2), f"{k}_dist1": npy(d1),
                    f"{k}_idx1": npy(i1), f"{k}_dist2": npy(d2), f"{k}_idx2": npy(i2)})
    # the reference's commented-out hand example (test_chamfer.py:42-43)
    x1 = torch.tensor([[[0, 0, 1], [1, 0, 0]]]).float()
    x2 = torch.tensor([[[0, 0, 1.1], [1.2, 0, 0]]]).float()
    d1, i1, d2, i2 = nn_distance_torch(x1, x2, "NWC")
    out.update({"hand_xyz1": npy(x1), "hand_xyz2": npy(x2), "hand_dist1": npy(d1),
                "hand_idx1": npy(i1), "hand_dist2": npy(d2), "hand_idx2": npy(i2)})
    # exact ties: points on a 4x4x4 lattice (torch.min on CPU returns the first minimum, which is
    # the lowest-index rule of chamfer_kernel.cu:82; asserted below against a sequential scan)
    x1 = (torch.randint(0, 4, (3, 200, 3), generator=g) * 0.25).float()
    x2 = (torch.randint(0, 4, (3, 300, 3), generator=g) * 0.25).float()
    d1, i1, d2, i2 = nn_distance_torch(x1, x2, "NWC")
    dm = ((x1[:, :, None] - x2[:, None]) ** 2).sum(-1)
    first1 = (dm == dm.min(2, keepdim=True)[0]).int().argmax(2)
    first2 = (dm == dm.min(1, keepdim=True)[0]).int().argmax(1)
    assert torch.equal(first1, i1) and torch.equal(first2, i2), "torch.min tie order changed"
    out.update({"tie_xyz1": npy(x1), "tie_xyz2": npy(x2), "tie_dist1": npy(d1),
                "tie_idx1": npy(i1), "tie_dist2": npy(d2), "tie_idx2": npy(i2)})
    # backward in float64 (the reference's gradcheck precision, test_chamfer.py:92-101):
    # autograd through min == ChamferBackwardKernel's formula (chamfer_kernel.cu:199-208)
    x1 = torch.rand(2, 64, 3, generator=g, dtype=torch.float64).requires_grad_()
    x2 = torch.rand(2, 48, 3, generator=g, dtype=torch.float64).requires_grad_()
    g1 = torch.randn(2, 64, generator=g, dtype=torch.float64)
    g2 = torch.randn(2, 48, generator=g, dtype=torch.float64)
    d1, i1, d2, i2 = nn_distance_torch(x1, x2, "NWC")
    ((d1 * g1).sum() + (d2 * g2).sum()).backward()
    out.update({"bwd_xyz1": npy(x1), "bwd_xyz2": npy(x2), "bwd_g1": npy(g1), "bwd_g2": npy(g2),
                "bwd_idx1": npy(i1), "bwd_idx2": npy(i2), "bwd_dist1": npy(d1),
                "bwd_dist2": npy(d2), "bwd_gxyz1": npy(x1.grad), "bwd_gxyz2": npy(x2.grad)})
    save("chamfer", **out)


# --------------------------------------------------------------------------------------------------
def gen_transforms(U):
    from scipy.spatial.transform import Rotation as R

    g = torch.Generator().manual_seed(1002)
    B, P, N = 2, 4, 64
    q = unit_quats(g, B, P)
    q[0, 3] = 0.0                      # padded part: zero quaternion -> identity (rotation.py:121-128)
    q[1, 2] = q[1, 2] * 0.3            # |q| = 0.3 <= 0.5 also becomes identity
    q[1, 3] = q[1, 3] * 1.7            # non-unit, kept as is: quaternion_apply scales by |q|^2
    t = torch.randn(B, P, 3, generator=g)
    pc = torch.randn(B, P, N, 3, generator=g)
    rot = U.Rotation3D(q, rot_type="quat")
    out = {"quat_in": npy(q), "quat_checked": npy(rot), "trans": npy(t), "pc": npy(pc),
           "rot_pc": npy(U.rot_pc(rot, pc)), "transform_pc": npy(U.transform_pc(t, rot, pc)),
           "qrot_flat": npy(U.qrot(rot.rot.reshape(-1, 4), pc[:, :, 0].reshape(-1, 3)))}
    # cross-check of the pytorch3d stand-in against scipy (unit quaternions only)
    qq = npy(rot)[0, :3].reshape(-1, 4)
    want = R.from_quat(qq[:, [1, 2, 3, 0]]).apply(npy(pc)[0, :3, 5])
    got = out["rot_pc"][0, :3, 5]
    assert np.abs(want - got).max() < 1e-5, np.abs(want - got).max()
    save("transforms", **out)


# --------------------------------------------------------------------------------------------------
def gen_losses(U):
    from multi_part_assembly.utils import loss as L

    g = torch.Generator().manual_seed(1003)
    B, P, N = 4, 5, 64
    pts = torch.randn(B, P, N, 3, generator=g) * 0.2
    valids = torch.tensor([[1, 0, 0, 0, 0], [1, 1, 0, 0, 0], [1, 1, 1, 1, 0], [1, 1, 1, 1, 1.0]])
    pts = pts * valids[..., None, None]
    q_gt = unit_quats(g, B, P) * valids[..., None]
    t_gt = torch.randn(B, P, 3, generator=g) * 0.3 * valids[..., None]
    q_pr = unit_quats(g, B, P).requires_grad_()
    t_pr = (torch.randn(B, P, 3, generator=g) * 0.3).requires_grad_()
    w = torch.rand(B, generator=g) + 0.5  # per-sample weights so every [B] entry gets a gradient
    out = {"pts": npy(pts), "valids": npy(valids), "quat_gt": npy(q_gt), "trans_gt": npy(t_gt),
           "quat_pred": npy(q_pr), "trans_pred": npy(t_pr), "w": npy(w)}

    def run(name, fn):
        for p in (q_pr, t_pr):
            p.grad = None
        r_pr = U.Rotation3D(q_pr, rot_type="quat")
        r_gt = U.Rotation3D(q_gt, rot_type="quat")
        res = fn(r_pr, r_gt)
        extra = ()
        if isinstance(res, tuple):
            res, *extra = res
        (res * w).sum().backward()
        out[name] = npy(res)
        out[name + "_gquat"] = npy(q_pr.grad) if q_pr.grad is not None else np.zeros((B, P, 4), "f4")
        out[name + "_gtrans"] = npy(t_pr.grad) if t_pr.grad is not None else np.zeros((B, P, 3), "f4")
        for i, e in enumerate(extra):
            out[f"{name}_pts{i + 1}"] = npy(e)

    run("trans_l2", lambda rp, rg: L.trans_l2_loss(t_pr, t_gt, valids))
    run("rot_cosine", lambda rp, rg: L.rot_cosine_loss(rp, rg, valids))
    run("rot_l2", lambda rp, rg: L.rot_l2_loss(rp, rg, valids))
    run("rot_points_l2", lambda rp, rg: L.rot_points_l2_loss(pts, rp, rg, valids))
    run("rot_points_cd", lambda rp, rg: L.rot_points_cd_loss(pts, rp, rg, valids, ret_pts=True))
    run("shape_cd_train", lambda rp, rg: L.shape_cd_loss(pts, t_pr, t_gt, rp, rg, valids,
                                                         ret_pts=True, training=True))
    run("shape_cd_eval", lambda rp, rg: L.shape_cd_loss(pts, t_pr, t_gt, rp, rg, valids,
                                                        ret_pts=False, training=False))
    save("losses", **out)


# --------------------------------------------------------------------------------------------------
def randomize_norm_params(model, g):
    """Non-trivial affine + running stats so that BN/LN parameters matter in the fixtures."""
    for m in model.modules():
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.LayerNorm)):
            with torch.no_grad():
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                if hasattr(m, "running_mean") and m.running_mean is not None:
                    m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                    m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)


def state_arrays(model, prefix="sd."):
    return {prefix + k: npy(v) for k, v in model.state_dict().items()}


def grad_arrays(model, prefix="grad."):
    return {prefix + k: npy(p.grad) for k, p in model.named_parameters()}


def gen_encoder(name, feat_dim, n, N, seed):
    from multi_part_assembly.models import build_encoder

    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    enc = build_encoder(name, feat_dim=feat_dim, global_feat=True)
    randomize_norm_params(enc, g)
    x = torch.randn(n, N, 3, generator=g) * 0.3
    w = torch.randn(n, feat_dim, generator=g)
    out = {"x": npy(x), "w": npy(w)}
    out.update(state_arrays(enc, "sd0."))          # state before the training-mode forward
    enc.train()
    xin = x.clone().requires_grad_()
    feat = enc(xin)
    (feat * w).sum().backward()
    out["feat_train"] = npy(feat)
    out["grad_x"] = npy(xin.grad)
    out.update(grad_arrays(enc))
    out.update(state_arrays(enc, "sd1."))          # running stats after one training step
    enc.eval()
    with torch.no_grad():
        out["feat_eval"] = npy(enc(x))
    if name == "dgcnn":
        from multi_part_assembly.models.modules.encoder import dgcnn as D
        with torch.no_grad():
            out["knn_idx_layer1"] = npy(D.knn(x.transpose(2, 1).contiguous(), k=20))
    save(name, **out)


def gen_dgcnn_graphs():
    """The kNN graph of EVERY EdgeConv stage as the reference builds it (dgcnn.py:8-15 called from :84-96): `knn` is
    wrapped while the reference's DGCNN runs a training-mode forward, and each call's input [n, C, N] (stored
    point-major [n, N, C]) and index output [n, N, 20] are recorded.  Case `a` repeats gen_encoder('dgcnn', 128, 3, 96,
    1005) draw for draw (same weights and cloud as dgcnn.npz); case `b` is a 2 x 1000-point cloud."""
    from multi_part_assembly.models import build_encoder
    from multi_part_assembly.models.modules.encoder import dgcnn as D

    out = {}
    for tag, (feat_dim, n, N, seed) in {"a": (128, 3, 96, 1005), "b": (128, 2, 1000, 1016)}.items():
        g = torch.Generator().manual_seed(seed)
        torch.manual_seed(seed)
        enc = build_encoder("dgcnn", feat_dim=feat_dim, global_feat=True)
        randomize_norm_params(enc, g)
        x = torch.randn(n, N, 3, generator=g) * 0.3
        calls = []
        real_knn = D.knn

        def spy(xx, k, _real=real_knn, _calls=calls):
            idx = _real(xx, k)
            _calls.append((npy(xx.transpose(2, 1).contiguous()), npy(idx)))
            return idx

        D.knn = spy
        try:
            enc.train()
            with torch.no_grad():
                feat = enc(x)
        finally:
            D.knn = real_knn
        assert len(calls) == 4 and [c[0].shape[-1] for c in calls] == [3, 64, 64, 128]
        out[f"{tag}.feat_train"] = npy(feat)
        for l, (xin, idx) in enumerate(calls):
            out[f"{tag}.stage{l + 1}.x"] = xin.astype(np.float32)
            out[f"{tag}.stage{l + 1}.idx"] = idx.astype(np.int16)
        if tag == "a":  # the same cloud and graph as dgcnn.npz
            ref = np.load(HERE / "dgcnn.npz")
            assert np.array_equal(ref["x"], calls[0][0]) and np.array_equal(ref["knn_idx_layer1"], calls[0][1])
            assert np.array_equal(ref["feat_train"], out["a.feat_train"])
        if tag == "b":  # the parameters of case b are not in any other fixture
            out.update(state_arrays(enc, "b.sd0."))
    save("dgcnn_graphs", **out)


def gen_transformer():
    from multi_part_assembly.models.pn_transformer.transformer import TransformerEncoder
    from multi_part_assembly.models.modules.regressor import StocasticPoseRegressor

    g = torch.Generator().manual_seed(1006)
    torch.manual_seed(1006)
    d, heads, ffn, layers = 64, 4, 128, 2
    enc = TransformerEncoder(d_model=d, num_heads=heads, ffn_dim=ffn, num_layers=layers,
                             norm_first=True, dropout=0.0)
    randomize_norm_params(enc, g)
    head = StocasticPoseRegressor(feat_dim=d, noise_dim=0, rot_type="quat")
    B, P = 3, 6
    tok = torch.randn(B, P, d, generator=g)
    valid = torch.tensor([[1, 1, 0, 0, 0, 0], [1, 1, 1, 1, 0, 0], [1, 1, 1, 1, 1, 1]]).bool()
    wr = torch.randn(B, P, 4, generator=g)
    wt = torch.randn(B, P, 3, generator=g)
    out = {"tokens": npy(tok), "valid": npy(valid), "w_rot": npy(wr), "w_trans": npy(wt),
           "cfg": np.array([d, heads, ffn, layers])}
    out.update(state_arrays(enc, "enc."))
    out.update(state_arrays(head, "head."))
    enc.train()
    head.train()
    tin = tok.clone().requires_grad_()
    feats = enc(tin, valid)
    rot, trans = head(feats)
    vm = valid[..., None].float()
    ((rot * wr * vm).sum() + (trans * wt * vm).sum()).backward()
    out.update({"feats": npy(feats), "rot": npy(rot), "trans": npy(trans), "grad_tokens": npy(tin.grad)})
    out.update(grad_arrays(enc, "genc."))
    out.update(grad_arrays(head, "ghead."))
    save("transformer", **out)


def synthetic_batch(g, B, P, N, num_parts):
    """Seeded stand-in for the data_dict contract (datasets/geometry_data.py:173-207)."""
    valids = torch.zeros(B, P)
    for b, k in enumerate(num_parts):
        valids[b, :k] = 1
    pcs = torch.randn(B, P, N, 3, generator=g) * 0.15
    pcs = pcs - pcs.mean(2, keepdim=True)
    pcs = pcs * valids[..., None, None]
    quat = unit_quats(g, B, P) * valids[..., None]
    trans = torch.randn(B, P, 3, generator=g) * 0.3 * valids[..., None]
    return {
        "part_pcs": pcs, "part_trans": trans, "part_quat": quat, "part_valids": valids,
        "part_label": torch.zeros(B, P, 0), "instance_label": torch.zeros(B, P, 0),
        "part_ids": torch.arange(P)[None].repeat(B, 1) * valids.long(),
        "valid_matrix": valids[:, :, None] * valids[:, None, :],
    }


def zero_dropout(model):
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
        if isinstance(m, nn.MultiheadAttention):
            m.dropout = 0.0


def gen_pn_transformer_step():
    from multi_part_assembly.models import build_model

    sys.path.insert(0, os.path.join(shim.REFERENCE_ROOT, "configs/pn_transformer/pn_transformer"))
    cfg = importlib.import_module("pn_transformer-32x1-cosine_400e-everyday").get_cfg_defaults()
    # shrink the widths so the fixture stays small; structure and loss config are the shipped ones
    cfg.model.pc_feat_dim = 64
    cfg.model.transformer_feat_dim = 128
    cfg.model.transformer_heads = 4
    cfg.model.transformer_layers = 2
    cfg.data.max_num_part = 5
    g = torch.Generator().manual_seed(1007)
    torch.manual_seed(1007)
    model = build_model(cfg)
    randomize_norm_params(model, g)
    zero_dropout(model)
    B, P, N = 3, 5, 64
    data = synthetic_batch(g, B, P, N, [2, 4, 5])
    out = {f"data.{k}": npy(v) for k, v in data.items()}
    out["cfg"] = np.array([64, 4, 128, 2])
    out.update(state_arrays(model, "sd0."))
    model.train()
    # forward_pass(mode='val') skips only the rank-0 logging block that needs a pl.Trainer
    # (base_model.py:137-146); self.training stays True so the training loss semantics apply.
    loss_dict = model.forward_pass({k: v.clone() for k, v in data.items()}, mode="val",
                                   optimizer_idx=-1)
    loss_dict["loss"].backward()
    for k, v in loss_dict.items():
        out[f"loss.{k}"] = npy(v)
    out.update(grad_arrays(model))
    out.update(state_arrays(model, "sd1."))
    # intermediate activations from a second, identical forward (BN in train mode uses batch stats,
    # so the values are the same; running stats are not re-saved)
    with torch.no_grad():
        pc_feats = model._extract_part_feats(data["part_pcs"], data["part_valids"])
        pred = model.forward({"part_pcs": data["part_pcs"], "part_valids": data["part_valids"],
                              "part_label": data["part_label"],
                              "instance_label": data["instance_label"]})
    out["act.pc_feats"] = npy(pc_feats)
    out["act.pred_rot"] = npy(pred["rot"])
    out["act.pred_trans"] = npy(pred["trans"])
    save("pn_transformer_step", **out)


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _load_cfg(rel_dir, module_name):
    sys.path.insert(0, os.path.join(shim.REFERENCE_ROOT, rel_dir))
    return importlib.import_module(module_name).get_cfg_defaults()


def _model_step(name, cfg, data, seed, extra=None):
    """One training-mode `forward_pass` + backward of a reference model built by `build_model(cfg)`.  The weights
    come from `param_fill.fill_parameters` (rebuilt from the names by the test, not stored); recorded: data, every
    loss term, parameter gradients and the buffers after the step (compact form for large tensors).
    `torch.manual_seed(seed + 1)` right before the pass fixes the draws the models make on the CPU generator
    (MoN noise, randperm of the matcher, GRU init)."""
    from multi_part_assembly.models import build_model
    import param_fill

    torch.manual_seed(seed)
    model = build_model(cfg)
    param_fill.fill_parameters(model, seed)
    zero_dropout(model)
    out = {f"data.{k}": npy(v) for k, v in data.items()}
    out["seed"] = np.array([seed])
    out["names"] = np.array(sorted(model.state_dict().keys()))
    out.update(extra or {})
    model.train()
    torch.manual_seed(seed + 1)
    loss_dict = model.forward_pass({k: v.clone() for k, v in data.items()}, mode="val", optimizer_idx=-1)
    loss_dict["loss"].backward()
    for k, v in loss_dict.items():
        if torch.is_tensor(v):
            out[f"loss.{k}"] = npy(v)
    for k, p in model.named_parameters():
        if p.grad is not None:
            out.update(param_fill.compact("grad.", k, npy(p.grad)))
    for k, v in model.state_dict().items():
        if "running_" in k:
            out.update(param_fill.compact("sd1.", k, npy(v)))
    # the same step once more in float64 (same weights, data and CPU-generator draws, all cast up): the anchor that
    # tells a test how far the float32 reference itself sits from the exact gradients of its own graph
    torch.manual_seed(seed)
    model64 = build_model(cfg)
    param_fill.fill_parameters(model64, seed)
    zero_dropout(model64)
    model64.double().train()
    # per BatchNorm module and channel: the smallest and the largest |output| of the float64 pass — the pre-activation of
    # the ReLU behind it.  A test that excuses an isolated ReLU flip of a float32 implementation must show that this
    # very channel has a pre-activation at the rounding level of zero (tests/test_callers_gpu.py, clause (c)).
    preact = {}

    def watch(mod_name):
        def hook(_m, _inp, outp):
            z = outp.detach().abs().transpose(0, 1).reshape(outp.shape[1], -1)
            lo, hi = z.min(dim=1).values, z.max(dim=1).values
            if mod_name in preact:
                lo, hi = torch.minimum(lo, preact[mod_name][0]), torch.maximum(hi, preact[mod_name][1])
            preact[mod_name] = (lo, hi)
        return hook

    for mod_name, mod in model64.named_modules():
        if isinstance(mod, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            mod.register_forward_hook(watch(mod_name))
    torch.manual_seed(seed + 1)  # the draws inside forward are float32 `.type_as(...)` upstream: the same values
    loss64 = model64.forward_pass({k: (v.double() if v.is_floating_point() else v.clone()) for k, v in data.items()},
                                  mode="val", optimizer_idx=-1)
    loss64["loss"].backward()
    for k, v in loss64.items():
        if torch.is_tensor(v):
            out[f"loss64.{k}"] = npy(v).astype(np.float64)
    for k, p in model64.named_parameters():
        if p.grad is not None:
            out.update(param_fill.compact("grad64.", k, npy(p.grad)))
    for mod_name, (lo, hi) in preact.items():
        out[f"preact64.{mod_name}"] = np.stack([npy(lo), npy(hi)]).astype(np.float64)
    save(name, **out)


def gen_dgl_step():
    """DGL on geometric data (configs/dgl/dgl-32x1-cosine_200e-everyday.py), widths shrunk."""
    cfg = _load_cfg("configs/dgl", "dgl-32x1-cosine_200e-everyday")
    cfg.model.pc_feat_dim = 64
    cfg.data.max_num_part = 5
    g = torch.Generator().manual_seed(1008)
    data = synthetic_batch(g, 3, 5, 64, [2, 4, 5])
    _model_step("dgl_step", cfg, data, 1008, {"cfg": np.array([64, 3])})


def gen_rgl_net_step():
    """RGL-NET on geometric data (configs/rgl_net/rgl_net-32x1-cosine_200e-everyday.py), widths shrunk."""
    cfg = _load_cfg("configs/rgl_net", "rgl_net-32x1-cosine_200e-everyday")
    cfg.model.pc_feat_dim = 64
    cfg.data.max_num_part = 5
    g = torch.Generator().manual_seed(1009)
    data = synthetic_batch(g, 3, 5, 64, [2, 4, 5])
    _model_step("rgl_net_step", cfg, data, 1009, {"cfg": np.array([64, 3])})


def gen_dgl_dgcnn_step():
    """BASELINE.json configs[2]: DGL with `cfg.model.encoder = 'dgcnn'` (models/dgl/network.py:90-99 feeding
    encoder/dgcnn.py:41-109), geometric data.  N = 64 points per part: k = 20 neighbours of 64."""
    cfg = _load_cfg("configs/dgl", "dgl-32x1-cosine_200e-everyday")
    cfg.model.encoder = "dgcnn"
    cfg.model.pc_feat_dim = 64
    cfg.data.max_num_part = 5
    g = torch.Generator().manual_seed(1014)
    data = synthetic_batch(g, 3, 5, 64, [2, 4, 5])
    _model_step("dgl_dgcnn_step", cfg, data, 1014, {"cfg": np.array([64, 3])})


def gen_rgl_net_dgcnn_artifact_step():
    """BASELINE.json configs[4]: RGL-NET with the DGCNN encoder on "artifact"-like data (many small parts: every
    sample has 4-5 of 5 parts, part extents a third of the everyday stand-in)."""
    cfg = _load_cfg("configs/rgl_net", "rgl_net-32x1-cosine_200e-everyday")
    cfg.model.encoder = "dgcnn"
    cfg.model.pc_feat_dim = 64
    cfg.data.max_num_part = 5
    g = torch.Generator().manual_seed(1015)
    data = synthetic_batch(g, 3, 5, 64, [4, 5, 5])
    data["part_pcs"] = data["part_pcs"] * 0.33
    _model_step("rgl_net_dgcnn_artifact_step", cfg, data, 1015, {"cfg": np.array([64, 3])})


def gen_lr_schedule():
    """utils/lr.py:26-125 CosineAnnealingWarmupRestarts exactly as base_model.py:407-425 builds it (cycle_mult and
    gamma 1, stepped once per epoch): the learning rate of every epoch for the shipped schedules."""
    from multi_part_assembly.utils.lr import CosineAnnealingWarmupRestarts

    out = {}
    for tag, (total, ratio, lr, decay) in {"pn400": (400, 0.05, 1e-3, 100.0), "dgl200": (200, 0.0, 1e-3, 100.0),
                                           "short": (10, 0.2, 5e-4, 10.0)}.items():
        p = nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([p], lr=lr)
        sched = CosineAnnealingWarmupRestarts(opt, total, max_lr=lr, min_lr=lr / decay,
                                              warmup_steps=int(total * ratio))
        lrs = []
        for _ in range(total + 5):  # a few epochs past the restart
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sched.step()
        out[f"{tag}.cfg"] = np.array([total, ratio, lr, decay])
        out[f"{tag}.lr"] = np.array(lrs, dtype=np.float64)
    save("lr_schedule", **out)


def gen_pn_refine_step():
    """PNTransformerRefine on geometric data (3 refinement rounds, loss summed over the rounds), widths shrunk."""
    cfg = _load_cfg("configs/pn_transformer/pn_transformer_refine", "pn_transformer_refine-32x1-cosine_400e-everyday")
    cfg.model.pc_feat_dim = 64
    cfg.model.transformer_pos_enc = (64, 64)
    cfg.model.transformer_feat_dim = 128
    cfg.model.transformer_heads = 4
    cfg.data.max_num_part = 5
    g = torch.Generator().manual_seed(1013)
    data = synthetic_batch(g, 3, 5, 64, [2, 4, 5])
    _model_step("pn_refine_step", cfg, data, 1013, {"cfg": np.array([64, 4, 128])})


def gen_global_semantic_step():
    """B-Global on semantic data (configs/global/global-32x1-cosine_200e-partnet_chair.py): Hungarian matching
    inside groups of identical parts + min-of-5 sampling with 32 noise channels (BASELINE.json configs[0])."""
    cfg = _load_cfg("configs/global", "global-32x1-cosine_200e-partnet_chair")
    cfg.model.pc_feat_dim = 64
    cfg.data.max_num_part = 5
    g = torch.Generator().manual_seed(1010)
    B, P, N = 2, 5, 128
    data = synthetic_batch(g, B, P, N, [4, 5])
    match_ids = torch.tensor([[0, 1, 1, 0, 0], [1, 1, 2, 2, 2]])
    # geometrically equivalent parts share one point cloud (what makes the matching meaningful)
    for b in range(B):
        for gid in range(1, int(match_ids[b].max()) + 1):
            members = torch.nonzero(match_ids[b] == gid).flatten().tolist()
            for m in members[1:]:
                data["part_pcs"][b, m] = data["part_pcs"][b, members[0]]
    data["match_ids"] = match_ids
    data["instance_label"] = torch.eye(P)[None].repeat(B, 1, 1) * data["part_valids"][..., None]
    _model_step("global_semantic_step", cfg, data, 1010, {"cfg": np.array([64])})


def gen_eval_metrics(U):
    """utils/eval_utils.py: part accuracy, translation / rotation metrics, connectivity accuracy; plus the
    evaluation-mode forward_pass of PNTransformer (eval-mode SCD, metrics merged into the loss dict)."""
    from multi_part_assembly.utils import eval_utils as E
    from multi_part_assembly.utils import Rotation3D

    g = torch.Generator().manual_seed(1011)
    B, P, N = 3, 5, 64
    data = synthetic_batch(g, B, P, N, [2, 4, 5])
    valids = data["part_valids"]
    gt_t, gt_q = data["part_trans"], data["part_quat"]
    # predictions: exact for some parts, slightly off for others, far off for the rest
    scale = torch.tensor([0.0, 0.003, 0.02, 0.3, 1.0])[None, :, None]
    pr_t = gt_t + scale * torch.randn(B, P, 3, generator=g) * valids[..., None]
    pr_q = torch.nn.functional.normalize(gt_q + scale * torch.randn(B, P, 4, generator=g), dim=-1) * valids[..., None]
    r_gt, r_pr = Rotation3D(gt_q.clone(), rot_type="quat"), Rotation3D(pr_q.clone(), rot_type="quat")
    out = {"pcs": npy(data["part_pcs"]), "valids": npy(valids), "gt_t": npy(gt_t), "gt_q": npy(gt_q),
           "pr_t": npy(pr_t), "pr_q": npy(pr_q)}
    out["part_acc"] = npy(E.calc_part_acc(data["part_pcs"], pr_t, gt_t, r_pr, r_gt, valids))
    for m in ("mse", "rmse", "mae"):
        out[f"trans_{m}"] = npy(E.trans_metrics(pr_t, gt_t, valids, m))
        out[f"rot_{m}"] = npy(E.rot_metrics(r_pr, r_gt, valids, m))
    out["euler_pr"] = npy(r_pr.to_euler(to_degree=True))
    contact = torch.zeros(B, P, P, 4)
    for b, k in enumerate([2, 4, 5]):
        for i in range(k - 1):  # a chain of contacts; the contact point sits between the two GT centroids
            mid = 0.5 * (gt_t[b, i] + gt_t[b, i + 1])
            for a, c in ((i, i + 1), (i + 1, i)):
                contact[b, a, c, 0] = 1.0
                # the contact point in part a's own frame: under the GT pose of a it lands on `mid`
                conj = gt_q[b, a] * torch.tensor([1.0, -1.0, -1.0, -1.0])
                contact[b, a, c, 1:] = U.qrot(conj, mid - gt_t[b, a]) + 0.004 * torch.randn(3, generator=g)
    out["contact_points"] = npy(contact)
    out["connectivity_acc_pred"] = npy(E.calc_connectivity_acc(pr_t, r_pr, contact))
    out["connectivity_acc_zero"] = npy(E.calc_connectivity_acc(torch.zeros_like(pr_t), Rotation3D(
        torch.tensor([1.0, 0, 0, 0]).repeat(B, P, 1), rot_type="quat"), contact))
    save("eval_metrics", **out)


def gen_pn_transformer_eval():
    """Evaluation-mode `forward_pass` of PNTransformer on the weights / data of the training-step fixture."""
    from multi_part_assembly.models import build_model

    sys.path.insert(0, os.path.join(shim.REFERENCE_ROOT, "configs/pn_transformer/pn_transformer"))
    cfg = importlib.import_module("pn_transformer-32x1-cosine_400e-everyday").get_cfg_defaults()
    cfg.model.pc_feat_dim = 64
    cfg.model.transformer_feat_dim = 128
    cfg.model.transformer_heads = 4
    cfg.model.transformer_layers = 2
    cfg.data.max_num_part = 5
    import param_fill
    torch.manual_seed(1012)
    model = build_model(cfg)
    param_fill.fill_parameters(model, 1012)
    g = torch.Generator().manual_seed(1012)
    data = synthetic_batch(g, 3, 5, 64, [2, 4, 5])
    model.eval()
    with torch.no_grad():
        res = model.forward_pass({k: v.clone() for k, v in data.items()}, mode="val", optimizer_idx=-1)
    out = {f"data.{k}": npy(v) for k, v in data.items()}
    out["seed"] = np.array([1012])
    out["cfg"] = np.array([64, 4, 128, 2])
    for k, v in res.items():
        out[f"res.{k}"] = npy(v) if torch.is_tensor(v) else np.array(v)
    save("pn_transformer_eval", **out)


# --------------------------------------------------------------------------------------------------
# --------------------------------------------------------------------------------------------------
# batch producers: the reference datasets' __getitem__ on seeded synthetic inputs
def build_partnet_mini(root):
    """A tiny dataset in the reference's PartNet on-disk format (written by this script, committed as data)."""
    g = np.random.default_rng(2024)
    (root / "shape_data").mkdir(parents=True, exist_ok=True)
    (root / "contact_points").mkdir(parents=True, exist_ok=True)
    specs = {101: [0, 4, 4, 4, 1, 2, 3], 102: [0, 1, 1, 2, 3, 4, 4, 4], 103: [3, 3, 1, 1], 104: list(range(9)),
             105: [2, 5]}
    np.save(root / "Chair.train.npy", np.array(list(specs), dtype=np.int64))
    for sid, geo in specs.items():
        p = len(geo)
        quat = g.normal(size=(p, 4))
        quat /= np.linalg.norm(quat, axis=1, keepdims=True)
        data = {
            "part_pcs": g.normal(size=(p, 32, 3)).astype(np.float32) * 0.1,
            "part_poses": np.concatenate([g.uniform(-0.5, 0.5, size=(p, 3)), quat], 1).astype(np.float32),
            "part_ids": g.integers(1, 6, size=p),
            "geo_part_ids": np.array(geo, dtype=np.int64),
            "sym": g.integers(0, 2, size=(p, 3)).astype(np.float32),
            "bbox": g.uniform(0.05, 0.3, size=(p, 3)).astype(np.float32),
        }
        np.save(root / "shape_data" / f"{sid}_level3.npy", data, allow_pickle=True)
        contact = np.zeros((p, p, 4), dtype=np.float32)
        contact[..., 0] = g.integers(0, 2, size=(p, p))
        contact[..., 1:] = g.uniform(-0.3, 0.3, size=(p, p, 3)) * contact[..., :1]
        np.save(root / "contact_points" / f"pairs_with_contact_points_{sid}_level3.npy", contact)


def gen_batch_producer():
    import random

    from multi_part_assembly.datasets.geometry_data import GeometryPartDataset
    from multi_part_assembly.datasets.partnet_data import PartNetPartDataset

    out = {}
    # geometry: __getitem__ with the mesh sampler replaced by seeded synthetic part clouds
    N, P = 64, 6
    g = np.random.default_rng(7)
    parts = [2, 5, 4]
    raws = [g.normal(size=(p, N, 3)) * g.uniform(0.02, 0.3, size=(p, 1, 3)) + g.uniform(-0.4, 0.4, size=(p, 1, 3))
            for p in parts]
    for tag, rot_range in (("free", -1), ("range", 30.0)):
        ds = object.__new__(GeometryPartDataset)
        ds.num_points, ds.min_num_part, ds.max_num_part = N, 2, P
        ds.shuffle_parts, ds.rot_range = False, rot_range
        ds.data_list = [f"item{i}" for i in range(len(parts))]
        ds.data_keys = ("part_ids", "valid_matrix")
        ds._get_pcs = lambda folder: raws[int(folder[4:])].copy()
        np.random.seed(77)
        random.seed(77)
        items = [ds[i] for i in range(len(parts))]
        for k in items[0]:
            out[f"geo.{tag}.{k}"] = np.stack([np.asarray(it[k]) for it in items])
    for i, r in enumerate(raws):
        out[f"geo.raw{i}"] = r
    # PartNet: the reference loader on the mini dataset
    root = HERE / "partnet_mini"
    build_partnet_mini(root)
    keys = ("part_label", "part_ids", "match_ids", "contact_points", "sym", "valid_matrix")
    ds = PartNetPartDataset(str(root), "Chair.train.npy", keys, num_part_category=5, min_num_part=2, max_num_part=8)
    out["partnet.shape_ids"] = np.array(ds.shape_ids)
    for i in range(len(ds)):
        for k, v in ds[i].items():
            out[f"partnet.{i}.{k}"] = np.asarray(v)
    save("batch_producer", **out)



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    only = set(filter(None, args.only.split(",")))
    shim.import_reference()
    import multi_part_assembly.utils as U

    todo = {
        "chamfer": gen_chamfer,
        "transforms": lambda: gen_transforms(U),
        "losses": lambda: gen_losses(U),
        "pointnet": lambda: gen_encoder("pointnet", 256, 5, 128, 1004),
        "dgcnn": lambda: gen_encoder("dgcnn", 128, 3, 96, 1005),
        "dgcnn_graphs": gen_dgcnn_graphs,
        "transformer": gen_transformer,
        "pn_transformer_step": gen_pn_transformer_step,
        "dgl_step": gen_dgl_step,
        "rgl_net_step": gen_rgl_net_step,
        "global_semantic_step": gen_global_semantic_step,
        "pn_refine_step": gen_pn_refine_step,
        "dgl_dgcnn_step": gen_dgl_dgcnn_step,
        "rgl_net_dgcnn_artifact_step": gen_rgl_net_dgcnn_artifact_step,
        "lr_schedule": gen_lr_schedule,
        "eval_metrics": lambda: gen_eval_metrics(U),
        "pn_transformer_eval": gen_pn_transformer_eval,
        "batch_producer": gen_batch_producer,
    }
    for name, fn in todo.items():
        if not only or name in only:
            fn()


if __name__ == "__main__":
    main()
