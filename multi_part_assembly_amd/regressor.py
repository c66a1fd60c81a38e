"""Pose head — mirror of the reference's PoseRegressor / StocasticPoseRegressor
(multi_part_assembly/models/modules/regressor.py:30-84); identical state_dict keys
(`fc_layers.{0,2}.*`, `rot_head.*`, `trans_head.*`).  Quaternion output only.

Compute: three small Linear layers over B*P <= 640 tokens — library GEMMs via PyTorch-ROCm.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class PoseRegressor(nn.Module):
    def __init__(self, feat_dim, rot_type="quat", norm_rot=True):
        super().__init__()
        if rot_type != "quat":
            raise NotImplementedError(f"rotation {rot_type} is not supported")
        self.rot_type, self.norm_rot = rot_type, norm_rot
        self.fc_layers = nn.Sequential(nn.Linear(feat_dim, 256), nn.LeakyReLU(0.2),
                                       nn.Linear(256, 128), nn.LeakyReLU(0.2))
        self.rot_head = nn.Linear(128, 4)
        self.trans_head = nn.Linear(128, 3)

    def forward(self, x):
        """x [B, C] or [B, P, C] -> (rot [.., 4] unit-normalised, trans [.., 3])."""
        hidden = self.fc_layers(x)
        rot = self.rot_head(hidden)
        if self.norm_rot:
            rot = F.normalize(rot, p=2, dim=-1)
        return rot, self.trans_head(hidden)


class StocasticPoseRegressor(PoseRegressor):
    """Appends `noise_dim` standard-normal channels to the input (MoN sampling); spelling as upstream."""

    def __init__(self, feat_dim, noise_dim, rot_type="quat", norm_rot=True):
        super().__init__(feat_dim + noise_dim, rot_type, norm_rot)
        self.noise_dim = noise_dim

    def forward(self, x):
        noise = torch.randn(*x.shape[:-1], self.noise_dim).type_as(x)
        return super().forward(torch.cat([x, noise], dim=-1))
