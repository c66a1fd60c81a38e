"""B-Global baseline — one PointNet per part plus one PointNet over the whole (untransformed) shape, concatenated
and fed to the pose head; mirror of the reference model (multi_part_assembly/models/b_global/network.py:7-132),
same sub-module names (`encoder`, `global_encoder`, `pose_predictor`) and state_dict keys.

Both encoders run on csrc/pointnet.hip: the part encoder through the mask-in / zeros-out entry, the global encoder
on the flattened [B, P*N, 3] cloud (20 000 points per shape at P = 20, N = 1000).  This is the model of the
reference's CPU-runnable plumbing configuration (BASELINE.json configs[0]: semantic data, Hungarian matching,
min-of-N sampling)."""
from __future__ import annotations

import torch

from .base_model import BaseModel
from .encoder import build_encoder
from .regressor import StocasticPoseRegressor


class GlobalModel(BaseModel):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.encoder = build_encoder(cfg.model.encoder, feat_dim=self.pc_feat_dim, global_feat=True)
        self.global_encoder = build_encoder(cfg.model.encoder, feat_dim=self.pc_feat_dim, global_feat=True)
        dim = 2 * self.pc_feat_dim
        if self.semantic:
            dim += self.max_num_part
        if self.use_part_label:
            dim += cfg.data.num_part_category
        self.pose_predictor = StocasticPoseRegressor(feat_dim=dim, noise_dim=cfg.loss.noise_dim,
                                                     rot_type=self.rot_type)

    def _extract_part_feats(self, part_pcs, part_valids):
        B, P, N, _ = part_pcs.shape
        return self.encoder.forward_parts(part_pcs.reshape(B * P, N, 3), part_valids.reshape(-1)).view(B, P, -1)

    def forward(self, data_dict):
        feats = data_dict.get("pre_pose_feats", None)
        if feats is None:
            part_pcs = data_dict["part_pcs"]
            pc_feats = self._extract_part_feats(part_pcs, data_dict["part_valids"])
            shape_feats = self.global_encoder(part_pcs.flatten(1, 2))             # [B, C], padded points included
            shape_feats = shape_feats[:, None].expand(-1, self.max_num_part, -1)
            feats = torch.cat([shape_feats, pc_feats, data_dict["part_label"].type_as(pc_feats),
                               data_dict["instance_label"].type_as(pc_feats)], dim=-1)
        rot, trans = self.pose_predictor(feats)
        return {"rot": self._wrap_rotation(rot), "trans": trans, "pre_pose_feats": feats}

    def _loss_function(self, data_dict, out_dict={}, optimizer_idx=-1):
        """One MoN sample; the features in front of the stochastic pose head are computed once and reused."""
        pred = self.forward({"part_pcs": data_dict["part_pcs"], "part_valids": data_dict["part_valids"],
                             "part_label": data_dict["part_label"], "instance_label": data_dict["instance_label"],
                             "pre_pose_feats": out_dict.get("pre_pose_feats", None)})
        loss_dict, new_out = self._calc_loss(pred, data_dict)
        new_out["pre_pose_feats"] = pred["pre_pose_feats"]
        return loss_dict, new_out
