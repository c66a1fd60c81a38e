"""PNTransformer — PointNet part encoder -> transformer over part tokens -> pose head; mirror of the
reference model (multi_part_assembly/models/pn_transformer/network.py:9-139), same sub-module names
(`encoder`, `corr_module`, `pose_predictor`) and therefore the same state_dict keys.

Valid-part compaction is done with fixed shapes: the encoder runs on the compacted [n, N, 3] valid
parts (BatchNorm statistics must only see valid parts, as upstream), but the gather/scatter index
is built on the device without the boolean-mask indexing of network.py:64-67.
"""
from __future__ import annotations

import torch

from .base_model import BaseModel
from .encoder import build_encoder
from .regressor import StocasticPoseRegressor
from .transformer import TransformerEncoder


class PNTransformer(BaseModel):
    def __init__(self, cfg):
        super().__init__(cfg)
        m = cfg.model
        self.encoder = build_encoder(m.encoder, feat_dim=self.pc_feat_dim, global_feat=True)
        self.corr_module = TransformerEncoder(
            d_model=self.pc_feat_dim, num_heads=m.transformer_heads, ffn_dim=m.transformer_feat_dim,
            num_layers=m.transformer_layers, norm_first=m.transformer_pre_ln)
        dim = self.pc_feat_dim
        if self.semantic:
            dim += self.max_num_part
        if self.use_part_label:
            dim += cfg.data.num_part_category
        self.pose_predictor = StocasticPoseRegressor(feat_dim=dim, noise_dim=cfg.loss.noise_dim,
                                                     rot_type=self.rot_type)

    def _extract_part_feats(self, part_pcs, part_valids):
        """[B, P, N, 3] -> [B, P, C]; padded slots get zeros (network.py:59-68)."""
        B, P, N, _ = part_pcs.shape
        if hasattr(self.encoder, "forward_parts"):  # HIP PointNet: mask in, zeros out, no host sync
            feats = self.encoder.forward_parts(part_pcs.reshape(B * P, N, 3), part_valids.reshape(-1))
            return feats.view(B, P, self.pc_feat_dim)
        valid = (part_valids == 1).reshape(-1)
        slots = torch.nonzero(valid, as_tuple=False).squeeze(1)          # [n] flat part indices
        feats = self.encoder(part_pcs.reshape(B * P, N, 3).index_select(0, slots))
        out = feats.new_zeros(B * P, self.pc_feat_dim)
        return out.index_copy(0, slots, feats).view(B, P, self.pc_feat_dim)

    def forward(self, data_dict):
        feats = data_dict.get("pre_pose_feats", None)
        if feats is None:
            part_valids = data_dict["part_valids"]
            pc_feats = self._extract_part_feats(data_dict["part_pcs"], part_valids)
            corr = self.corr_module(pc_feats, part_valids == 1)
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


def build_model(cfg):
    """Registry of reference models/__init__.py:10-26 (the LSTM / identity baselines are out of scope)."""
    if cfg.model.name == "pn_transformer":
        return PNTransformer(cfg)
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
