"""PNTransformer — PointNet part encoder -> transformer over part tokens -> pose head; mirror of the
reference model (multi_part_assembly/models/pn_transformer/network.py:9-139), same sub-module names
(`encoder`, `corr_module`, `pose_predictor`) and therefore the same state_dict keys.

Valid-part compaction is done with fixed shapes: the encoder runs on the compacted [n, N, 3] valid
parts (BatchNorm statistics must only see valid parts, as upstream), but the gather/scatter index
is built on the device without the boolean-mask indexing of network.py:64-67.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .base_model import BaseModel
from .encoder import build_encoder
from .regressor import StocasticPoseRegressor
from .transformer import TransformerEncoder


class PNTransformer(BaseModel):
    def __init__(self, cfg):
        super().__init__(cfg)
        m = cfg.model
        self.encoder = build_encoder(m.encoder, feat_dim=self.pc_feat_dim, global_feat=True)
        self.corr_module = self._init_corr_module()
        self.pose_predictor = self._init_pose_predictor()

    def _init_corr_module(self):
        m = self.cfg.model
        return TransformerEncoder(d_model=self.pc_feat_dim, num_heads=m.transformer_heads,
                                  ffn_dim=m.transformer_feat_dim, num_layers=m.transformer_layers,
                                  norm_first=m.transformer_pre_ln)

    def _init_pose_predictor(self):
        dim = self.pc_feat_dim
        if self.semantic:
            dim += self.max_num_part
        if self.use_part_label:
            dim += self.cfg.data.num_part_category
        return StocasticPoseRegressor(feat_dim=dim, noise_dim=self.cfg.loss.noise_dim, rot_type=self.rot_type)

    def _extract_part_feats(self, part_pcs, part_valids):
        """[B, P, N, 3] -> [B, P, C]; padded slots get zeros (network.py:59-68)."""
        B, P, N, _ = part_pcs.shape
        # mask in, zeros out: the valid parts are counted and compacted on the device, no host sync
        feats = self.encoder.forward_parts(part_pcs.reshape(B * P, N, 3), part_valids.reshape(-1))
        return feats.view(B, P, self.pc_feat_dim)

    def forward(self, data_dict):
        feats = data_dict.get("pre_pose_feats", None)
        if feats is None:
            part_valids = data_dict["part_valids"]
            pc_feats = self._extract_part_feats(data_dict["part_pcs"], part_valids)
            corr = self.corr_module(pc_feats, part_valids)  # (real iff == 1: applied inside the attention kernels)
            feats = torch.cat([corr, data_dict["part_label"].type_as(corr),
                               data_dict["instance_label"].type_as(corr)], dim=-1)
        rot, trans = self.pose_predictor(feats)
        return {"rot": self._wrap_rotation(rot), "trans": trans, "pre_pose_feats": feats}

    def _loss_function(self, data_dict, out_dict={}, optimizer_idx=-1):
        """One MoN sample: predict, then `_calc_loss`; the encoder/transformer features are cached in
        `pre_pose_feats` because only the pose head is stochastic (network.py:106-139)."""
        pred = self.forward({
            "part_pcs": data_dict["part_pcs"], "part_valids": data_dict["part_valids"],
            "part_label": data_dict["part_label"], "instance_label": data_dict["instance_label"],
            "pre_pose_feats": out_dict.get("pre_pose_feats", None)})
        loss_dict, new_out = self._calc_loss(pred, data_dict)
        new_out["pre_pose_feats"] = pred["pre_pose_feats"]
        return loss_dict, new_out


class PosEncoder(nn.Module):
    """MLP positional encoding of a pose vector (network_refine.py:11-25); `layers.{0,2,..}` as upstream."""

    def __init__(self, dims):
        super().__init__()
        layers = []
        for i in range(len(dims) - 2):
            layers += [nn.Linear(dims[i], dims[i + 1]), nn.ReLU()]
        layers.append(nn.Linear(dims[-2], dims[-1]))
        self.layers = nn.Sequential(*layers)

    def forward(self, x):
        return self.layers(x)


class PNTransformerRefine(PNTransformer):
    """PNTransformer with iterative refinement (pn_transformer/network_refine.py:28-175): `refine_steps` rounds of
    [pose -> positional encoding, added to the part tokens -> that round's transformer -> that round's pose head];
    the loss is summed over the rounds.  Same sub-module names (`corr_module.{i}`, `pose_predictor.{i}`,
    `corr_pos_enc.layers.*`).  Each round's transformer and every loss evaluation run on the HIP path; the pose
    heads' odd input widths (features + 7 pose values) use the library-op fallback of `PoseRegressor`."""

    def __init__(self, cfg):
        self.refine_steps = cfg.model.refine_steps
        self.pose_pc_feat = cfg.model.pose_pc_feat
        super().__init__(cfg)
        zero_pose = torch.zeros(1, 1, self.pose_dim)
        zero_pose[..., 0] = 1.0
        self.register_buffer("zero_pose", zero_pose, persistent=False)  # on the module's device: graph-capturable
        self.corr_pos_enc = PosEncoder([self.pose_dim, *cfg.model.transformer_pos_enc])

    def _init_corr_module(self):
        m = self.cfg.model
        return nn.ModuleList(
            TransformerEncoder(d_model=self.pc_feat_dim, num_heads=m.transformer_heads, ffn_dim=m.transformer_feat_dim,
                               num_layers=m.transformer_layers, norm_first=m.transformer_pre_ln,
                               out_dim=self.pc_feat_dim) for _ in range(self.refine_steps))

    def _init_pose_predictor(self):
        dim = self.pc_feat_dim + self.pose_dim
        if self.semantic:
            dim += self.max_num_part
        if self.pose_pc_feat:
            dim += self.pc_feat_dim
        if self.use_part_label:
            dim += self.cfg.data.num_part_category
        return nn.ModuleList(
            StocasticPoseRegressor(feat_dim=dim, noise_dim=self.cfg.loss.noise_dim, rot_type=self.rot_type)
            for _ in range(self.refine_steps))

    def forward(self, data_dict):
        pc_feats = data_dict.get("pc_feats", None)
        part_valids = data_dict["part_valids"]
        if pc_feats is None:
            pc_feats = self._extract_part_feats(data_dict["part_pcs"], part_valids)
        part_label = data_dict["part_label"].type_as(pc_feats)
        inst_label = data_dict["instance_label"].type_as(pc_feats)
        B, P = inst_label.shape[:2]
        pose = self.zero_pose.to(pc_feats).expand(B, P, -1)
        valid_mask = part_valids  # (real iff == 1: applied inside the attention kernels)
        tokens, rots, transs = pc_feats, [], []
        for i in range(self.refine_steps):
            tokens = self.corr_module[i](tokens + self.corr_pos_enc(pose), valid_mask)
            feats = torch.cat([tokens, part_label, inst_label, pose], dim=-1)
            if self.pose_pc_feat:
                feats = torch.cat([pc_feats, feats], dim=-1)
            rot, trans = self.pose_predictor[i](feats)
            rots.append(rot)
            transs.append(trans)
            pose = torch.cat([rot, trans], dim=-1)
        if self.training:
            rot, trans = self._wrap_rotation(torch.stack(rots, dim=0)), torch.stack(transs, dim=0)
        else:
            rot, trans = self._wrap_rotation(rots[-1]), transs[-1]
        return {"rot": rot, "trans": trans, "pc_feats": pc_feats}

    def _loss_function(self, data_dict, out_dict={}, optimizer_idx=-1):
        pred = self.forward({"part_pcs": data_dict["part_pcs"], "part_valids": data_dict["part_valids"],
                             "part_label": data_dict["part_label"], "instance_label": data_dict["instance_label"],
                             "pc_feats": out_dict.get("pc_feats", None)})
        if not self.training:
            loss_dict, out = self._calc_loss(pred, data_dict)
            out["pc_feats"] = pred["pc_feats"]
            return loss_dict, out
        total, out = None, {}
        for i in range(self.refine_steps):
            loss_dict, out = self._calc_loss({"rot": pred["rot"][i], "trans": pred["trans"][i]}, data_dict)
            if total is None:
                total = {k: 0.0 for k in loss_dict}
            for k, v in loss_dict.items():
                total[k] = total[k] + v
                total[f"{k}_{i}"] = v
        out["pc_feats"] = pred["pc_feats"]
        return total, out


def build_model(cfg):
    """Registry of reference models/__init__.py:10-26 (the LSTM / identity baselines are out of scope)."""
    model = _build_model(cfg)
    TransformerEncoder.assign_dropout_salts(model)  # sibling encoders draw different masks, reproducibly
    return model


def _build_model(cfg):
    if cfg.model.name == "pn_transformer":
        return PNTransformer(cfg)
    if cfg.model.name == "pn_transformer_refine":
        return PNTransformerRefine(cfg)
    if cfg.model.name == "global":
        from .global_model import GlobalModel
        return GlobalModel(cfg)
    if cfg.model.name == "dgl":
        from .gnn import DGLModel
        return DGLModel(cfg)
    if cfg.model.name == "rgl_net":
        from .gnn import RGLNet
        return RGLNet(cfg)
    raise NotImplementedError(f"Model {cfg.model.name} not supported")
