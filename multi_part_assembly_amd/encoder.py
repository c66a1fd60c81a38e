"""Per-part point-cloud encoders — mirrors of the reference's PointNet and DGCNN
(multi_part_assembly/models/modules/encoder/pointnet.py:6-41, dgcnn.py:8-109) with identical
constructor arguments, forward contract ([n, N, 3] -> [n, feat_dim]) and state_dict keys, so a
reference checkpoint loads unchanged (DGCNN registers every BatchNorm under two names, `bnK` and
`convK.1`, exactly as upstream).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class PointNet(nn.Module):
    """Shared MLP 3-64-64-64-128-F (1x1 conv, no bias, BN, ReLU except after the last) + max over N."""

    WIDTHS = (3, 64, 64, 64, 128)

    def __init__(self, feat_dim, global_feat=True):
        super().__init__()
        dims = (*self.WIDTHS, feat_dim)
        for i in range(5):
            setattr(self, f"conv{i + 1}", nn.Conv1d(dims[i], dims[i + 1], kernel_size=1, bias=False))
        for i in range(5):
            setattr(self, f"bn{i + 1}", nn.BatchNorm1d(dims[i + 1]))
        self.global_feat = global_feat

    def forward(self, x):
        h = x.transpose(2, 1).contiguous()  # [n, 3, N]
        for i in range(1, 6):
            h = getattr(self, f"bn{i}")(getattr(self, f"conv{i}")(h))
            if i < 5:
                h = F.relu(h)
        return h.max(dim=-1)[0] if self.global_feat else h.transpose(2, 1).contiguous()


def knn(x, k):
    """Indices [n, N, k] of the k nearest points (self included) in feature space; x [n, C, N].

    Same Gram-form scores as the reference (dgcnn.py:8-15) so the neighbour SETS agree; the order
    inside k is irrelevant downstream (max over k)."""
    inner = torch.matmul(x.transpose(2, 1), x)
    sq = (x * x).sum(dim=1, keepdim=True)
    score = -sq - (-2 * inner) - sq.transpose(2, 1)
    return score.topk(k=k, dim=-1)[1]


def get_graph_feature(x, k=20):
    """Edge features [x_j - x_i ; x_i] for the kNN graph: x [n, C, N] -> [n, 2C, N, k]."""
    n, C, N = x.shape
    idx = knn(x, k)
    pts = x.transpose(2, 1)                                            # [n, N, C]
    nbr = torch.gather(pts[:, None].expand(n, N, N, C), 2, idx[..., None].expand(n, N, k, C))
    ctr = pts[:, :, None].expand(n, N, k, C)
    return torch.cat((nbr - ctr, ctr), dim=3).permute(0, 3, 1, 2).contiguous()


class DGCNN(nn.Module):
    """4 EdgeConv stages (k=20, widths 64-64-128-256, LeakyReLU 0.2, max over k), concat 512 ->
    1x1 conv -> [max ; mean] over N -> Linear."""

    def __init__(self, feat_dim, global_feat=True):
        super().__init__()
        self.bn1, self.bn2 = nn.BatchNorm2d(64), nn.BatchNorm2d(64)
        self.bn3, self.bn4 = nn.BatchNorm2d(128), nn.BatchNorm2d(256)
        self.bn5 = nn.BatchNorm1d(feat_dim)
        act = lambda: nn.LeakyReLU(negative_slope=0.2)
        self.conv1 = nn.Sequential(nn.Conv2d(6, 64, kernel_size=1, bias=False), self.bn1, act())
        self.conv2 = nn.Sequential(nn.Conv2d(128, 64, kernel_size=1, bias=False), self.bn2, act())
        self.conv3 = nn.Sequential(nn.Conv2d(128, 128, kernel_size=1, bias=False), self.bn3, act())
        self.conv4 = nn.Sequential(nn.Conv2d(256, 256, kernel_size=1, bias=False), self.bn4, act())
        self.conv5 = nn.Sequential(nn.Conv1d(512, feat_dim, kernel_size=1, bias=False), self.bn5, act())
        self.global_feat = global_feat
        if global_feat:
            self.out_fc = nn.Linear(feat_dim * 2, feat_dim)

    def forward(self, x):
        h = x.transpose(2, 1).contiguous()
        stages = []
        for conv in (self.conv1, self.conv2, self.conv3, self.conv4):
            h = conv(get_graph_feature(h)).max(dim=-1)[0]
            stages.append(h)
        h = self.conv5(torch.cat(stages, dim=1))
        if not self.global_feat:
            return h.transpose(2, 1).contiguous()
        return self.out_fc(torch.cat((h.max(dim=-1)[0], h.mean(dim=-1)), dim=1))


def build_encoder(arch, feat_dim, global_feat=True, **kwargs):
    """Registry of reference encoder/__init__.py:6-21 restricted to the hot-path encoders."""
    if arch == "pointnet":
        return PointNet(feat_dim, global_feat=global_feat)
    if arch == "dgcnn":
        return DGCNN(feat_dim, global_feat=global_feat)
    raise NotImplementedError(f"{arch} is not supported (PointNet++ is outside the hot path)")
