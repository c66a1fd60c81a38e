"""CPU restatement of the hot path's other callers — DGL (models/dgl/network.py:14-297, dgl/modules.py:5-86), RGL-NET
(models/rgl_net/network.py:14-162, rgl_net/modules.py:5-30, modules/rnn.py:6-46) and B-Global
(models/b_global/network.py:7-132) with the loss assembly of models/modules/base_model.py:150-238,240-387
(GT <-> prediction matching inside groups of identical parts, min-of-N sampling).  Stock torch ops on state-dict
tensors, like oracle/nets.py.  TEST INFRASTRUCTURE: used by tests/ (pinned by the reference's own forward_pass fixtures,
tests/test_oracle_golden.py) and by bench.py's `cpu_baseline` leg for configs c1 / c3 / c5."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import geometry as og
from . import nets as on


def _lin(x, sd, name):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def pair_mlp(x, sd, prefix, training, stats_out, final_relu=True):
    """MLP3 / MLP4 of dgl/modules.py:5-58: x [R, L, cin] -> [R, L, F]; three Conv1d(k=1) + BatchNorm1d (+ ReLU)."""
    h = x.transpose(1, 2)
    for i in (1, 2, 3):
        h = F.conv1d(h, sd[f"{prefix}conv{i}.weight"], sd[f"{prefix}conv{i}.bias"])
        h = on._bn(h, sd, f"{prefix}bn{i}", training, stats_out)
        if i < 3 or final_relu:
            h = F.relu(h)
    return h.transpose(1, 2)


def relation_net(x, sd, prefix):
    """dgl/modules.py:61-73."""
    return torch.sigmoid(_lin(F.relu(_lin(F.relu(_lin(x, sd, prefix + "mlp1")), sd, prefix + "mlp2")), sd, prefix + "mlp3"))


def pose_encoder(x, sd, prefix):
    """dgl/modules.py:76-86."""
    return F.relu(_lin(F.relu(_lin(x, sd, prefix + "mlp1")), sd, prefix + "mlp2"))


def pose_head_noise(x, sd, prefix, noise_dim):
    """StocasticPoseRegressor (modules/regressor.py:71-84): `noise_dim` standard-normal channels appended."""
    if noise_dim:
        x = torch.cat([x, torch.randn(*x.shape[:-1], noise_dim).type_as(x)], dim=-1)
    return on.pose_head(x, sd, prefix)


def part_features(sd, batch, encoder, prefix, training, stats_out):
    """_extract_part_feats (dgl/network.py:90-99, b_global/network.py:45-54): valid parts through the encoder, zeros else."""
    pcs, valids = batch["part_pcs"], batch["part_valids"]
    B, P = valids.shape
    mask = valids == 1
    enc = on.pointnet if encoder == "pointnet" else on.dgcnn
    feats = enc(pcs[mask], sd, prefix, training, stats_out)
    return torch.zeros(B, P, feats.shape[-1], dtype=feats.dtype).index_put((mask,), feats)


def packed_bigru(x, hidden, valids, sd, prefix, training):
    """RNNWrapper(nn.GRU(bidirectional, batch_first)) of modules/rnn.py:6-46 on state-dict tensors: pack by
    valids.sum(1), run the GRU on the packed data exactly as nn.GRU.forward does for a PackedSequence (initial state
    permuted by the packing's sort order), pad back to P steps (zeros behind a sample's last valid part)."""
    from torch.nn.utils.rnn import PackedSequence, pack_padded_sequence, pad_packed_sequence
    packed = pack_padded_sequence(x, valids.sum(dim=1).cpu(), batch_first=True, enforce_sorted=False)
    params = [sd[prefix + n + sfx] for sfx in ("", "_reverse") for n in ("rnn.weight_ih_l0", "rnn.weight_hh_l0",
                                                                          "rnn.bias_ih_l0", "rnn.bias_hh_l0")]
    hx = hidden.index_select(1, packed.sorted_indices)
    data, _ = torch._VF.gru(packed.data, packed.batch_sizes, hx, params, True, 1, 0.0, training, True)
    out, _ = pad_packed_sequence(PackedSequence(data, packed.batch_sizes, packed.sorted_indices, packed.unsorted_indices),
                                 batch_first=True, total_length=x.shape[1])
    return out


def dgl_forward(sd, batch, iters, encoder="dgcnn", training=True, stats_out=None, recurrent=False, merge_node=False):
    """DGL.forward for geometric data (dgl/network.py:154-243; no semantic labels, merge_node without effect):
    -> list of (rot [B,P,4], trans [B,P,3]) per GNN iteration.
    recurrent: RGLNet.forward (rgl_net/network.py:70-162) — the node update runs [part_feats ; messages] through a
    packed bidirectional GRU with a random initial state (two draws per iteration on the CPU generator, :50-57) and
    MLP4 without the last ReLU (rgl_net/modules.py:5-30) instead of DGL's MLP4 on [messages ; part_feats].
    merge_node (cfg.model.merge_node; on in the RGL-NET everyday config): odd iterations predict the relations with
    `relation_predictor` instead of `relation_predictor_dense` (dgl/network.py:127-132) — the node merging itself needs
    semantic labels and does nothing on geometric data (:195)."""
    part_feats = part_features(sd, batch, encoder, "encoder.", training, stats_out)
    valid_matrix = batch["valid_matrix"]
    B, P, Fd = part_feats.shape
    pose = torch.zeros(B, P, 7, dtype=part_feats.dtype)
    pose[..., 0] = 1.0
    preds = []
    for it in range(iters):
        if it == 0:
            relation = valid_matrix
        else:
            pf = pose_encoder(pose, sd, "pose_extractor.")
            pair = torch.cat([pf[:, None].expand(B, P, P, -1), pf[:, :, None].expand(B, P, P, -1)], dim=-1)
            which = "relation_predictor." if merge_node and it % 2 == 1 else "relation_predictor_dense."
            relation = relation_net(pair.reshape(B, P * P, -1), sd, which).view(B, P, P) * valid_matrix
        pair = torch.cat([part_feats[:, :, None].expand(B, P, P, Fd), part_feats[:, None].expand(B, P, P, Fd)], dim=-1)
        edge = pair_mlp(pair.reshape(B * P, P, 2 * Fd), sd, f"edge_mlps.{it}.", training, stats_out).view(B, P, P, -1)
        msg = (edge * relation[..., None]).sum(dim=2) / (relation.sum(dim=-1, keepdim=True) + 1e-6)
        if recurrent:
            hidden = torch.cat([torch.randn((1, B, Fd)).repeat(2, 1, 1), torch.randn((2, B, Fd))], dim=-1).type_as(msg)
            gru_out = packed_bigru(torch.cat([part_feats, msg], dim=-1), hidden, batch["part_valids"], sd, f"grus.{it}.",
                                   training)
            part_feats = pair_mlp(gru_out, sd, f"node_mlps.{it}.", training, stats_out, final_relu=False)
        else:
            part_feats = pair_mlp(torch.cat([msg, part_feats], dim=-1), sd, f"node_mlps.{it}.", training, stats_out)
        rot, trans = on.pose_head(torch.cat([part_feats, pose], dim=-1), sd, f"pose_predictors.{it}.")
        pose = torch.cat([rot, trans], dim=-1)
        preds.append((rot, trans))
    return preds


def dgl_loss(sd, batch, iters, encoder="dgcnn", training=True, stats_out=None, recurrent=False, merge_node=False):
    """forward_pass of DGL (recurrent: RGL-NET) on geometric data: the loss of every iteration's prediction, summed
    (dgl/network.py:245-297, inherited by RGLNet)."""
    preds = dgl_forward(sd, batch, iters, encoder, training, stats_out, recurrent, merge_node)
    total = {}
    for i, (rot, trans) in enumerate(preds):
        terms = og.calc_loss_geometric(og.checked_quat(rot), trans, batch["part_pcs"], og.checked_quat(batch["part_quat"]),
                                       batch["part_trans"], batch["part_valids"], training=training)
        for k, v in terms.items():
            total[k] = total.get(k, 0.0) + v.mean()
            total[f"{k}_{i}"] = v.mean()
    return total


@torch.no_grad()
def match_parts(part_pcs, pred_trans, pred_quat, gt_trans, gt_quat, match_ids):
    """base_model.py:150-238: inside every group of geometrically equivalent parts, permute the GT poses so that they
    line up with the predictions at minimum Chamfer cost (100 sub-sampled points, scipy's Hungarian solver)."""
    from scipy.optimize import linear_sum_assignment
    B, P, N, _ = part_pcs.shape
    new_t, new_q = gt_trans.clone(), gt_quat.clone()
    ids = match_ids.long()
    for b in range(B):
        for g in range(1, int(ids[b].max()) + 1):
            idx = torch.nonzero(ids[b] == g).flatten()
            if idx.numel() == 0:
                continue
            p = idx.numel()
            sample = torch.randperm(N)[:min(100, N)]
            pts = part_pcs[b, idx][:, sample]
            p1 = og.transform_pc(pred_trans[b, idx][None], pred_quat[b, idx][None], pts[None])[0]
            p2 = og.transform_pc(gt_trans[b, idx][None], gt_quat[b, idx][None], pts[None])[0]
            a = p1[:, None].expand(p, p, -1, 3).reshape(p * p, -1, 3)
            c = p2[None].expand(p, p, -1, 3).reshape(p * p, -1, 3)
            d1, d2 = og.chamfer_distance(a, c)
            rind, cind = linear_sum_assignment((d1.mean(1) + d2.mean(1)).view(p, p).numpy())
            new_t[b, idx[rind]] = gt_trans[b, idx[cind]]
            new_q[b, idx[rind]] = gt_quat[b, idx[cind]]
    return new_t, new_q


def semantic_loss_terms(pred_quat, pred_trans, batch, loss_cfg):
    """_calc_loss on a semantic dataset (base_model.py:240-314): matching first, then the five terms, each [B]; the
    whole-shape term uses the training normalisation."""
    pcs, valids = batch["part_pcs"], batch["part_valids"]
    gt_q = og.checked_quat(batch["part_quat"])
    new_t, new_q = match_parts(pcs, pred_trans.detach(), pred_quat.detach(), batch["part_trans"], gt_q, batch["match_ids"])
    out = {"trans_loss": og.trans_l2_loss(pred_trans, new_t, valids),
           "rot_pt_cd_loss": og.rot_points_cd_loss(pcs, pred_quat, new_q, valids),
           "transform_pt_cd_loss": og.shape_cd_loss(pcs, pred_trans, new_t, pred_quat, new_q, valids, training=True)}
    if loss_cfg.get("use_rot_loss", False):
        out["rot_loss"] = og.rot_cosine_loss(pred_quat, new_q, valids)
    if loss_cfg.get("use_rot_pt_l2_loss", False):
        out["rot_pt_l2_loss"] = og.rot_points_l2_loss(pcs, pred_quat, new_q, valids)
    return out


def global_loss(sd, batch, loss_cfg, sample_iter, noise_dim, encoder="pointnet", training=True, stats_out=None):
    """forward_pass of B-Global on semantic data (b_global/network.py:56-132 + base_model.py:348-387): part and
    whole-shape features once, `sample_iter` stochastic pose predictions, per sample the one with the smallest total."""
    pcs = batch["part_pcs"]
    B, P = batch["part_valids"].shape
    pc_feats = part_features(sd, batch, encoder, "encoder.", training, stats_out)
    enc = on.pointnet if encoder == "pointnet" else on.dgcnn
    shape_feats = enc(pcs.flatten(1, 2), sd, "global_encoder.", training, stats_out)
    feats = torch.cat([shape_feats[:, None].expand(-1, P, -1), pc_feats, batch["part_label"].type_as(pc_feats),
                       batch["instance_label"].type_as(pc_feats)], dim=-1)
    samples = {}
    for _ in range(sample_iter):
        rot, trans = pose_head_noise(feats, sd, "pose_predictor.", noise_dim)
        terms = semantic_loss_terms(og.checked_quat(rot), trans, batch, loss_cfg)
        for k, v in terms.items():
            samples.setdefault(k, []).append(v)
    stacked = {k: torch.stack(v, dim=0) for k, v in samples.items()}
    total = 0.0
    for k, v in stacked.items():
        total = total + v * loss_cfg[f"{k}_w"]
    stacked["loss"] = total
    best = total.argmin(0)
    cols = torch.arange(B)
    return {k: v[best, cols].mean() for k, v in stacked.items()}
