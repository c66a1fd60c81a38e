"""Graph-network callers of the hot path: DGL (dynamic graph learning) and RGL-NET (DGL + bidirectional GRU) —
mirrors of the reference models (multi_part_assembly/models/dgl/network.py:14-297, dgl/modules.py:5-86,
rgl_net/network.py:12-162, rgl_net/modules.py:5-30, modules/rnn.py:6-46) with the same sub-module names, hence
the same state_dict keys (`edge_mlps.{i}.conv1.weight`, `node_mlps.{i}.bn3.running_mean`, `pose_predictors.{i}.*`,
`relation_predictor_dense.mlp1.*`, `pose_extractor.*`, `grus.{i}.rnn.weight_ih_l0`, ...).

What runs where: the part encoder (PointNet: csrc/pointnet.hip, DGCNN: csrc/dgcnn_enc.hip), every loss evaluation of
the `gnn_iter` stacked predictions (fused assembly loss: csrc/assembly_loss.hip, grid_nn.hip), the optimiser, and the
bulk of the graph network — the P x P edge MLPs, the node MLPs and the wide layers of the relation nets (Conv1d /
Linear + BatchNorm1d + ReLU layers: csrc/mlp.hip, exact-fp32 MFMA), the recurrence of RGL-NET's bidirectional GRU
(csrc/gru.hip: all steps of both directions in one launch) and the pose heads — are the HIP hot path, as are the small pieces between them (the 7-wide
first layer of the pose encoder, the 512 -> 1 relation head with its sigmoid and mask, the relation-weighted mean:
csrc/gnn_glue.hip); the GRU's input projection and the concatenations stay on PyTorch-ROCm library ops.

Differences from the reference, none of them numerical beyond fp32 re-association:
  * part features are extracted with the mask-in / zeros-out PointNet entry (no boolean-mask sync);
  * the GRU runs on the padded batch and the padded steps are masked instead of packing sequences through a
    `lengths.cpu()` round trip (modules/rnn.py:28) — see `_MaskedBiGRU`.
"""
from __future__ import annotations

import copy

import numpy as np
import torch
import torch.nn as nn

from .base_model import BaseModel
from .encoder import build_encoder
from .gnn_ops import (NARROW_MAX_IN, RELATION_MEAN_MAX_PARTS, narrow_linear_relu, pair_rows, pair_rows_supported,
                      relation_head, relation_head_supported, relation_mean)
from .gru import gru_recurrent, supported as gru_supported
from .loss import LossTerms
from .mlp import bn_counter_batch, mlp_layer, pair_layer, pair_layer_supported, supported as mlp_supported
from .regressor import StocasticPoseRegressor


class _PairMLP(nn.Module):
    """conv1 (cin -> 512) - bn - relu - conv2 (512 -> 512) - bn - relu - conv3 (512 -> F) - bn [- relu]; the three
    1x1 Conv1d + BatchNorm1d of dgl/modules.py:5-58 (`MLP3` / `MLP4`, identical there) and rgl_net/modules.py:5-30."""

    def __init__(self, in_dim, feat_len, final_relu=True):
        super().__init__()
        self.conv1 = nn.Conv1d(in_dim, 512, 1)
        self.conv2 = nn.Conv1d(512, 512, 1)
        self.conv3 = nn.Conv1d(512, feat_len, 1)
        self.bn1 = nn.BatchNorm1d(512)
        self.bn2 = nn.BatchNorm1d(512)
        self.bn3 = nn.BatchNorm1d(feat_len)
        self.final_relu = final_relu

    def _rows(self, x):
        """x [rows, cin] -> [rows, F]: the three Conv1d(k=1) + BatchNorm1d (+ ReLU) layers on the HIP library
        (csrc/mlp.hip: exact-fp32 MFMA GEMMs, fixed-order BatchNorm statistics over all rows — BatchNorm1d over the
        (batch, position) axes of the reference's [R, C, L] layout is exactly that)."""
        h = mlp_layer(x, self.conv1.weight, self.conv1.bias, self.bn1, relu=True, training=self.training)
        h = mlp_layer(h, self.conv2.weight, self.conv2.bias, self.bn2, relu=True, training=self.training)
        return mlp_layer(h, self.conv3.weight, self.conv3.bias, self.bn3, relu=self.final_relu, training=self.training)

    PAIR_LAYER = True  # (a knob for A/B runs and tests: False = the first layer as a GEMM over the materialised pair rows)
    MIN_ROWS = 1  # (a knob for A/B runs: tools/probe_node_mlp.py — the HIP layers win from the 640-row node MLP up, and
                  # unlike the library's split-K weight gradients they are deterministic)

    def _hip_ok(self, x, cin, rows):
        return (x.is_cuda and rows >= self.MIN_ROWS and mlp_supported(cin, 512)
                and mlp_supported(512, self.conv3.out_channels))

    def _tail(self, h):
        """h [R, 512, L]: output of conv1 -> [R, L, F] (library ops: widths outside csrc/mlp.hip)."""
        h = torch.relu(self.bn1(h))
        h = torch.relu(self.bn2(self.conv2(h)))
        h = self.bn3(self.conv3(h))
        if self.final_relu:
            h = torch.relu(h)
        return h.transpose(1, 2)

    def forward(self, x):
        """x [R, L, cin] -> [R, L, F]."""
        R, L, cin = x.shape
        if self._hip_ok(x, cin, R * L):
            return self._rows(x.reshape(R * L, cin)).view(R, L, -1)
        return self._tail(self.conv1(x.transpose(1, 2)))

    def forward_pairs(self, a, b):
        """Rows = (sample, part i), positions = part j, input [a_i ; b_j]: a, b [B, P, F] -> [B*P, P, F_out]."""
        B, P, F = a.shape
        if (self._hip_ok(a, 2 * F, B * P * P) and self.PAIR_LAYER and pair_layer_supported(F, 512)
                and P <= 1024 and B * P <= (1 << 20)):  # (mpa_pair_layer_*'s envelope; beyond it: the pair rows below)
            # conv1 of [a_i ; b_j] = (a Wa^T + bias)_i + (b Wb^T)_j: two GEMMs over the B*P part rows, the pair tensor is
            # never built (csrc/mlp.hip: mpa_pair_layer_*); BatchNorm statistics over all B*P*P rows as upstream
            h = pair_layer(a, b, self.conv1.weight, self.conv1.bias, self.bn1, relu=True, training=self.training)
            h = mlp_layer(h, self.conv2.weight, self.conv2.bias, self.bn2, relu=True, training=self.training)
            h = mlp_layer(h, self.conv3.weight, self.conv3.bias, self.bn3, relu=self.final_relu, training=self.training)
            return h.view(B * P, P, -1)
        if self._hip_ok(a, 2 * F, B * P * P):
            if pair_rows_supported(F):
                pair = pair_rows(a, b)  # one launch; its backward sums the two halves over j / over i in one more
            else:
                pair = torch.cat([a[:, :, None, :].expand(B, P, P, F), b[:, None, :, :].expand(B, P, P, F)], dim=-1)
            return self._rows(pair.reshape(B * P * P, 2 * F)).view(B * P, P, -1)
        # library ops: the first conv applied to the two halves separately, the pair tensor is never materialised
        w = self.conv1.weight[:, :, 0]                         # [512, 2F]
        pa = a @ w[:, :F].t() + self.conv1.bias                # [B, P, 512]  (depends on i)
        pb = b @ w[:, F:].t()                                  # [B, P, 512]  (depends on j)
        h = pa[:, :, None, :] + pb[:, None, :, :]              # [B, P_i, P_j, 512]
        return self._tail(h.reshape(B * P, P, 512).transpose(1, 2))


class RelationNet(nn.Module):
    """Pair of pose features -> relation weight in (0, 1) (dgl/modules.py:61-73)."""

    def __init__(self):
        super().__init__()
        self.mlp1 = nn.Linear(128 + 128, 256)
        self.mlp2 = nn.Linear(256, 512)
        self.mlp3 = nn.Linear(512, 1)

    def forward(self, x, mask=None):
        """x [B, P*P, 256] -> [B, P*P, 1] (times `mask` [B, P*P], if given); the two wide layers on csrc/mlp.hip, the
        512 -> 1 head with its sigmoid and the mask in one launch of csrc/gnn_glue.hip."""
        if (x.is_cuda and x.numel() // x.shape[-1] >= _PairMLP.MIN_ROWS and mlp_supported(x.shape[-1], 256)
                and mlp_supported(256, 512) and relation_head_supported(512)):
            lead = x.shape[:-1]
            h = mlp_layer(x.reshape(-1, x.shape[-1]), self.mlp1.weight, self.mlp1.bias, None, relu=True)
            h = mlp_layer(h, self.mlp2.weight, self.mlp2.bias, None, relu=True)
            m = None if mask is None else mask.reshape(-1)
            return relation_head(h, self.mlp3.weight, self.mlp3.bias, m).view(*lead, 1)
        out = torch.sigmoid(self.mlp3(torch.relu(self.mlp2(torch.relu(self.mlp1(x))))))
        return out if mask is None else out * mask.reshape(*out.shape)


class PoseEncoder(nn.Module):
    """Pose vector -> 128-d feature (dgl/modules.py:76-86)."""

    def __init__(self, pose_dim):
        super().__init__()
        self.mlp1 = nn.Linear(pose_dim, 256)
        self.mlp2 = nn.Linear(256, 128)

    def forward(self, x):
        if x.is_cuda and x.shape[-1] <= NARROW_MAX_IN and mlp_supported(256, 128):
            # the 7-wide layer in one launch of csrc/gnn_glue.hip, the 256 -> 128 layer on csrc/mlp.hip
            h = narrow_linear_relu(x, self.mlp1.weight, self.mlp1.bias)
            lead = h.shape[:-1]
            return mlp_layer(h.reshape(-1, 256), self.mlp2.weight, self.mlp2.bias, None, relu=True).view(*lead, 128)
        return torch.relu(self.mlp2(torch.relu(self.mlp1(x))))


def _clones(module, n):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(n)])


class DGLModel(BaseModel):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.iter = cfg.model.gnn_iter
        self.merge_node = cfg.model.merge_node
        zero_pose = torch.zeros(1, 1, self.pose_dim)
        zero_pose[..., 0] = 1.0  # identity quaternion, zero translation (base_model.py:31-34)
        # a plain attribute upstream; a non-persistent buffer here (same state_dict keys) so that it lives on the
        # module's device: a host-to-device copy inside forward cannot be captured into a HIP graph
        self.register_buffer("zero_pose", zero_pose, persistent=False)
        self.encoder = build_encoder(cfg.model.encoder, feat_dim=self.pc_feat_dim, global_feat=True)
        self.edge_mlps = _clones(_PairMLP(2 * self.pc_feat_dim, self.pc_feat_dim), self.iter)
        self.node_mlps = self._init_node_mlps()
        dim = self.pc_feat_dim + self.pose_dim
        if self.semantic:
            dim += self.max_num_part
        if self.use_part_label:
            dim += cfg.data.num_part_category
        self.pose_predictors = _clones(
            StocasticPoseRegressor(feat_dim=dim, noise_dim=cfg.loss.noise_dim, rot_type=self.rot_type), self.iter)
        self.relation_predictor_dense = RelationNet()
        if self.merge_node:
            self.relation_predictor = RelationNet()
        self.pose_extractor = PoseEncoder(self.pose_dim)

    def _init_node_mlps(self):
        return _clones(_PairMLP(2 * self.pc_feat_dim, self.pc_feat_dim), self.iter)

    # ---- pieces of one GNN iteration -----------------------------------------------------------------------
    def _extract_part_feats(self, part_pcs, part_valids):
        B, P, N, _ = part_pcs.shape
        return self.encoder.forward_parts(part_pcs.reshape(B * P, N, 3), part_valids.reshape(-1)).view(B, P, -1)

    def _gather_same_class(self, data_dict):
        """Index groups of geometrically equivalent parts per sample (dgl/network.py:75-88); host-side, once per
        step, semantic datasets only."""
        class_list = data_dict.get("class_list", None)
        if self.merge_node and self.semantic and class_list is None:
            valids, ids = data_dict["part_valids"].cpu().numpy(), data_dict["part_ids"].cpu().numpy()
            class_list = []
            for v, i in zip(valids, ids):
                real = i[v == 1]
                class_list.append([np.where(real == lbl)[0] for lbl in np.unique(real)])
        return class_list

    @staticmethod
    def _merge_nodes(part_feats, pose_feats, class_list):
        """Equivalent parts share the channel-wise max of their features (dgl/network.py:101-119)."""
        part_out, pose_out = part_feats.clone(), pose_feats.clone()
        for b, groups in enumerate(class_list):
            for idx in groups:
                if len(idx) > 1:
                    part_out[b, idx] = part_feats[b, idx].max(dim=-2, keepdim=True)[0]
                    pose_out[b, idx] = pose_feats[b, idx].max(dim=-2, keepdim=True)[0]
        return part_out, pose_out

    def _update_relation(self, pose_feats, iter_ind, mask=None):
        """relation[b, i, j] = net([pose_j ; pose_i]) (dgl/network.py:121-133), times `mask` [B, P, P] if given (the
        valid matrix the caller multiplies by, dgl/network.py:213)."""
        B, P, C = pose_feats.shape
        if pose_feats.is_cuda and pair_rows_supported(C):  # row (i, j) = [pose_j ; pose_i]
            pair = pair_rows(pose_feats, pose_feats, swap=True)
        else:
            pair = torch.cat([pose_feats[:, None].expand(B, P, P, C), pose_feats[:, :, None].expand(B, P, P, C)], dim=-1)
        net = self.relation_predictor if (self.merge_node and iter_ind % 2 == 1) else self.relation_predictor_dense
        m = None if mask is None else mask.reshape(B, P * P)
        return net(pair.reshape(B, P * P, 2 * C), m).view(B, P, P)

    def _message_passing(self, part_feats, relation, iter_ind):
        """Relation-weighted mean of the edge features (dgl/network.py:135-152)."""
        B, P, _ = part_feats.shape
        edge = self.edge_mlps[iter_ind].forward_pairs(part_feats, part_feats).view(B, P, P, -1)
        if edge.is_cuda and P <= RELATION_MEAN_MAX_PARTS:
            return relation_mean(edge, relation.expand(B, P, P))  # one launch of csrc/gnn_glue.hip
        msg = (edge * relation[..., None]).sum(dim=2)
        return msg / (relation.sum(dim=-1, keepdim=True) + 1e-6)

    def _node_update(self, part_feats, messages, data_dict, iter_ind):
        return self.node_mlps[iter_ind](torch.cat([messages, part_feats], dim=-1))

    def forward(self, data_dict):
        with bn_counter_batch():  # (the BatchNorm step counters of all MLP layers in one launch)
            return self._forward(data_dict)

    def _forward(self, data_dict):
        # per-forward scratch (RGL-NET keeps its GRU index plan here): a shallow copy, so that nothing is left in — or
        # later trusted from — the caller's batch dict, whose tensors a loader may refill in place
        data_dict = dict(data_dict)
        part_feats = data_dict.get("part_feats", None)
        if part_feats is None:
            part_feats = self._extract_part_feats(data_dict["part_pcs"], data_dict["part_valids"])
        local_feats = part_feats
        valid_matrix = data_dict["valid_matrix"]
        part_label = data_dict["part_label"].type_as(part_feats)
        instance_label = data_dict["instance_label"].type_as(part_feats)
        B, P = instance_label.shape[:2]
        pred_pose = self.zero_pose.to(part_feats).expand(B, P, -1)
        class_list = self._gather_same_class(data_dict)
        rots, transs = [], []
        for it in range(self.iter):
            if it == 0:
                feats_in, relation = part_feats, valid_matrix  # fully connected start
            else:
                pose_feats = self.pose_extractor(pred_pose)
                feats_in = part_feats
                if self.merge_node and self.semantic and it % 2 == 1:
                    feats_in, pose_feats = self._merge_nodes(part_feats, pose_feats, class_list)
                relation = self._update_relation(pose_feats, it, valid_matrix)
            messages = self._message_passing(feats_in, relation, it)
            part_feats = self._node_update(part_feats, messages.type_as(part_feats), data_dict, it)
            rot, trans = self.pose_predictors[it](torch.cat([part_feats, part_label, instance_label, pred_pose], dim=-1))
            pred_pose = torch.cat([rot, trans], dim=-1)
            rots.append(rot)
            transs.append(trans)
        if self.training:
            rot, trans = self._wrap_rotation(torch.stack(rots, dim=0)), torch.stack(transs, dim=0)
        else:
            rot, trans = self._wrap_rotation(rots[-1]), transs[-1]
        out = {"rot": rot, "trans": trans, "part_feats": local_feats, "class_list": class_list}
        if self.training:
            # the per-iteration predictions as they were produced: the loss reads these, so that autograd does not
            # scatter each iteration's gradient into a zero-filled stack and add the stacks up again
            out["_iters"] = [(self._wrap_rotation(r), t) for r, t in zip(rots, transs)]
        return out

    def _loss_function(self, data_dict, out_dict={}, optimizer_idx=-1):
        """Loss of every iteration's prediction, summed (dgl/network.py:245-297); `part_feats` / `class_list` are
        reused across MoN samples."""
        pred = self.forward({k: data_dict[k] for k in ("part_pcs", "part_valids", "part_label", "instance_label",
                                                      "part_ids", "valid_matrix")}
                            | {"part_feats": out_dict.get("part_feats", None),
                               "class_list": out_dict.get("class_list", None)})
        keep = {"part_feats": pred["part_feats"], "class_list": pred["class_list"]}
        if not self.training:
            loss_dict, out = self._calc_loss(pred, data_dict)
            out.update(keep)
            return loss_dict, out
        per_iter, out = [], {}
        iters = pred.get("_iters") or [(pred["rot"][i], pred["trans"][i]) for i in range(self.iter)]
        for rot, trans in iters:
            loss_dict, out = self._calc_loss({"rot": rot, "trans": trans}, data_dict)
            per_iter.append(loss_dict)
        out.update(keep)
        names = list(per_iter[0])
        stacks = [getattr(d, "stacked", None) for d in per_iter]
        if all(st is not None and list(st[0]) == names for st in stacks):
            # the fused loss hands every iteration's terms over as ONE [K, B] tensor: sum those (iter - 1 launches), and
            # let `loss_function` weight the [K (iter + 1), B] concatenation directly — summing and re-stacking the
            # terms one by one was ~100 small launches per step, forward and backward
            summed = stacks[0][1]
            for st in stacks[1:]:
                summed = summed + st[1]
            full = torch.cat([summed] + [st[1] for st in stacks], dim=0)
            keys = names + [f"{k}_{i}" for i in range(self.iter) for k in names]
            total = LossTerms((k, full[j]) for j, k in enumerate(keys))
            total.stacked = (tuple(keys), full)
            return total, out
        total = {k: 0.0 for k in names}
        for i, loss_dict in enumerate(per_iter):
            for k, v in loss_dict.items():
                total[k] = total[k] + v
                total[f"{k}_{i}"] = v
        return total, out


class _MaskedBiGRU(nn.Module):
    """`RNNWrapper(nn.GRU(bidirectional))` of the reference (modules/rnn.py:6-46) with the same `.rnn` sub-module
    (state_dict keys) but without `pack_padded_sequence`: the reference packs by `valids.sum(1).cpu()` — a device
    sync per iteration.  Valid parts come first in every sample, so the forward direction over the padded batch
    already equals the packed run on the valid steps; the reverse direction is run on the per-sample REVERSED valid
    prefix (a gather built on the device).  Padded steps output zeros, as `pad_packed_sequence` does."""

    def __init__(self, rnn, batch_first=True):
        super().__init__()
        assert batch_first and rnn.bidirectional and rnn.num_layers == 1
        self.rnn = rnn
        self.batch_first = batch_first

    @staticmethod
    def plan(valids, T):
        """(rev_idx [B, T] int64, mask [B, T, 1] float) of a validity matrix: what every GRU call on the same batch
        needs — the caller may build it once per step and hand it to each call (`plan=`)."""
        B = valids.shape[0]
        lengths = valids.sum(dim=1).long()                                    # [B], stays on the device
        steps = torch.arange(T, device=valids.device)[None].expand(B, T)
        inside = steps < lengths[:, None]
        return torch.where(inside, lengths[:, None] - 1 - steps, steps), inside[..., None].to(torch.float32)

    def forward(self, x, hidden=None, valids=None, plan=None):
        if valids is None:
            return self.rnn(x, hidden)
        B, T, _ = x.shape
        H = self.rnn.hidden_size
        rev_idx, mask = plan if plan is not None else self.plan(valids, T)
        x_rev = torch.gather(x, 1, rev_idx[..., None].expand_as(x))
        w = self.rnn
        if x.is_cuda and gru_supported(H, B):
            # both directions' T steps in ONE launch (csrc/gru.hip); the two input projections are plain GEMMs
            gi = torch.stack([torch.nn.functional.linear(x, w.weight_ih_l0, w.bias_ih_l0),
                              torch.nn.functional.linear(x_rev, w.weight_ih_l0_reverse, w.bias_ih_l0_reverse)])
            hs = gru_recurrent(gi, hidden[0:2], torch.stack([w.weight_hh_l0, w.weight_hh_l0_reverse]),
                               torch.stack([w.bias_hh_l0, w.bias_hh_l0_reverse]))
            fwd, bwd = hs[0], hs[1]
        else:
            fwd = torch._VF.gru(x, hidden[0:1].contiguous(),
                                [w.weight_ih_l0, w.weight_hh_l0, w.bias_ih_l0, w.bias_hh_l0],
                                True, 1, 0.0, self.training, False, True)[0]
            bwd = torch._VF.gru(x_rev, hidden[1:2].contiguous(),
                                [w.weight_ih_l0_reverse, w.weight_hh_l0_reverse, w.bias_ih_l0_reverse,
                                 w.bias_hh_l0_reverse], True, 1, 0.0, self.training, False, True)[0]
        bwd = torch.gather(bwd, 1, rev_idx[..., None].expand(B, T, H))       # back to part order
        out = torch.cat([fwd, bwd], dim=-1) * mask.to(x.dtype)
        return out, None


class RGLNet(DGLModel):
    def __init__(self, cfg):
        super().__init__(cfg)
        gru = nn.GRU(input_size=2 * self.pc_feat_dim, hidden_size=2 * self.pc_feat_dim, num_layers=1, batch_first=True,
                     dropout=0, bidirectional=True)
        self.grus = _clones(_MaskedBiGRU(gru), self.iter)

    def _init_node_mlps(self):
        # consumes the GRU's 4F outputs; no ReLU after the last BatchNorm (rgl_net/modules.py:24-28)
        return _clones(_PairMLP(4 * self.pc_feat_dim, self.pc_feat_dim, final_relu=False), self.iter)

    def _init_gru_hidden(self, B, device=None):
        """Same random draws, in the same order, as rgl_net/network.py:50-57 (CPU generator).  While a HIP graph is
        being captured the draws come from the device generator instead (a host-to-device copy cannot be a graph
        node; torch advances the captured generator's offset on every replay)."""
        cuda = device is not None and device.type == "cuda"
        if cuda and torch.cuda.is_current_stream_capturing():
            rand_vec = torch.randn((1, B, self.pc_feat_dim), device=device).repeat(2, 1, 1)
            zero_vec = torch.randn((2, B, self.pc_feat_dim), device=device)
            return torch.cat([rand_vec, zero_vec], dim=-1)
        # the draws land in pinned memory and travel with an asynchronous copy: a copy from pageable memory makes the
        # host wait until the stream has drained (three pipeline drains per RGL-NET step, one per GNN iteration)
        rand_vec = torch.randn((1, B, self.pc_feat_dim), pin_memory=cuda)
        zero_vec = torch.randn((2, B, self.pc_feat_dim), pin_memory=cuda)
        if cuda:
            rand_vec, zero_vec = rand_vec.to(device, non_blocking=True), zero_vec.to(device, non_blocking=True)
        return torch.cat([rand_vec.repeat(2, 1, 1), zero_vec], dim=-1)

    def _node_update(self, part_feats, messages, data_dict, iter_ind):
        hidden = self._init_gru_hidden(part_feats.shape[0], messages.device).type_as(messages)
        valids = data_dict["part_valids"]
        plan = data_dict.get("_gru_plan")  # `_forward`'s own scratch dict: one index plan for the GRU calls of ONE forward
        if plan is None:
            plan = data_dict["_gru_plan"] = _MaskedBiGRU.plan(valids, part_feats.shape[1])
        gru_out, _ = self.grus[iter_ind](torch.cat([part_feats, messages], dim=-1), hidden, valids=valids, plan=plan)
        return self.node_mlps[iter_ind](gru_out)
