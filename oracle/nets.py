"""CPU restatement (stock torch functional ops, fp32) of the network half of the training step:
PointNet / DGCNN encoders, the pre-LN transformer over part tokens, the pose head and the full
PNTransformer loss.  TEST INFRASTRUCTURE — see oracle/__init__.py.  Also the "reference-equivalent
PyTorch CPU path" that bench.py times as `cpu_baseline` (the reference's own Python cannot travel
to the GPU box and has no CPU Chamfer).

Parameters come in as a flat dict keyed like the reference's state_dict, so fixtures captured from
the reference load directly.  Pinned by tests/test_oracle_golden.py against tests/golden/
{pointnet,dgcnn,transformer,pn_transformer_step}.npz.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import geometry as og

BN_EPS, BN_MOMENTUM, LN_EPS = 1e-5, 0.1, 1e-5


def _bn(x, sd, name, training, stats_out=None):
    """nn.BatchNorm{1,2}d: batch statistics (biased var) in training, running stats in eval; the
    running-stat update (momentum 0.1, unbiased var) is written to `stats_out` if given."""
    rm, rv = sd[name + ".running_mean"], sd[name + ".running_var"]
    if training and stats_out is not None:
        rm, rv = rm.clone(), rv.clone()
        y = F.batch_norm(x, rm, rv, sd[name + ".weight"], sd[name + ".bias"], True, BN_MOMENTUM, BN_EPS)
        stats_out[name + ".running_mean"], stats_out[name + ".running_var"] = rm, rv
        return y
    return F.batch_norm(x, None if training else rm, None if training else rv, sd[name + ".weight"],
                        sd[name + ".bias"], training, BN_MOMENTUM, BN_EPS)


def pointnet(x, sd, prefix="", training=True, stats_out=None):
    """models/modules/encoder/pointnet.py:29-41 (global_feat=True): x [n, N, 3] -> [n, F]."""
    h = x.transpose(2, 1).contiguous()
    for i in range(1, 6):
        h = _bn(F.conv1d(h, sd[f"{prefix}conv{i}.weight"]), sd, f"{prefix}bn{i}", training, stats_out)
        if i < 5:
            h = F.relu(h)
    return h.max(dim=-1)[0]


def knn_indices(x, k):
    """models/modules/encoder/dgcnn.py:8-15: top-k of the negated Gram-form squared distances."""
    inner = -2 * torch.matmul(x.transpose(2, 1), x)
    xx = torch.sum(x ** 2, dim=1, keepdim=True)
    return (-xx - inner - xx.transpose(2, 1)).topk(k=k, dim=-1)[1]


def graph_feature(x, k=20, idx=None):
    """dgcnn.py:18-38: [n, C, N] -> [n, 2C, N, k] = [x_j - x_i ; x_i].  `idx` [n, N, k] (tests only): a given graph
    instead of the search — lets a float64 evaluation run on the float32 reference's graphs."""
    n, C, N = x.shape
    idx = (knn_indices(x, k) if idx is None else idx.long()) + torch.arange(n).view(-1, 1, 1) * N
    pts = x.transpose(2, 1).contiguous()
    nbr = pts.view(n * N, C)[idx.view(-1)].view(n, N, k, C)
    ctr = pts.view(n, N, 1, C).repeat(1, 1, k, 1)
    return torch.cat((nbr - ctr, ctr), dim=3).permute(0, 3, 1, 2).contiguous()


def dgcnn(x, sd, prefix="", training=True, stats_out=None, graphs=None):
    """dgcnn.py:77-109 (global_feat=True): x [n, N, 3] -> [n, F].  `graphs`: optional list of 4 index tensors."""
    h = x.transpose(2, 1).contiguous()
    stages = []
    for i in range(1, 5):
        e = F.conv2d(graph_feature(h, idx=None if graphs is None else graphs[i - 1]), sd[f"{prefix}conv{i}.0.weight"])
        e = F.leaky_relu(_bn(e, sd, f"{prefix}bn{i}", training, stats_out), 0.2)
        h = e.max(dim=-1)[0]
        stages.append(h)
    h = F.conv1d(torch.cat(stages, dim=1), sd[f"{prefix}conv5.0.weight"])
    h = F.leaky_relu(_bn(h, sd, f"{prefix}bn5", training, stats_out), 0.2)
    pooled = torch.cat((h.max(dim=-1)[0], h.mean(dim=-1)), dim=1)
    return F.linear(pooled, sd[f"{prefix}out_fc.weight"], sd[f"{prefix}out_fc.bias"])


def dropout_keep_scale(seed, site, numel, p):
    """Keep-scale (0 or 1/(1-p)) of the `numel` elements of dropout site `site`: numpy restatement of the
    counter-based generator of csrc/transformer.hip (splitmix64 finaliser of seed + golden*(site+1) + index,
    top 24 bits as a uniform in [0, 1), dropped when u < p).  Test infrastructure only: it lets the oracle
    apply the SAME masks as the HIP kernels, so training-mode dropout is checked value by value."""
    import numpy as np
    m64 = (1 << 64) - 1
    base = (int(seed) + 0x9E3779B97F4A7C15 * (int(site) + 1)) & m64
    with np.errstate(over="ignore"):
        x = np.uint64(base) + np.arange(numel, dtype=np.uint64)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    u = (x >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return torch.from_numpy(np.where(u < np.float32(p), np.float32(0.0), scale).astype(np.float32))


def transformer_encoder(tokens, valid, sd, prefix, num_layers, num_heads, dropout_p=0.0, seed=0):
    """models/pn_transformer/transformer.py:63-79 with norm_first=True:
    x += drop(MHA(LN1(x))) ; x += drop(W2 drop(relu(W1 LN2(x)))) per layer, final LayerNorm; padded keys
    masked.  dropout_p = 0 (the default) disables the 4 dropout sites of nn.TransformerEncoderLayer; otherwise
    their masks come from `dropout_keep_scale` (sites 4l .. 4l+3: attention probabilities, attention output,
    FFN hidden, FFN output)."""
    x = tokens
    B, P, D = x.shape
    hd = D // num_heads

    def drop(t, site):
        if dropout_p <= 0.0:
            return t
        return t * dropout_keep_scale(seed, site, t.numel(), dropout_p).view(t.shape)

    neg = torch.zeros(B, 1, 1, P).masked_fill(~valid[:, None, None, :], float("-inf"))
    for l in range(num_layers):
        p = f"{prefix}transformer_encoder.layers.{l}."
        h = F.layer_norm(x, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], LN_EPS)
        qkv = F.linear(h, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"])
        q, k, v = (t.view(B, P, num_heads, hd).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
        att = drop(torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd) + neg, dim=-1), 4 * l)
        h = (att @ v).transpose(1, 2).reshape(B, P, D)
        h = F.linear(h, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        x = x + drop(h, 4 * l + 1)
        h = F.layer_norm(x, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], LN_EPS)
        h = drop(F.relu(F.linear(h, sd[p + "linear1.weight"], sd[p + "linear1.bias"])), 4 * l + 2)
        h = F.linear(h, sd[p + "linear2.weight"], sd[p + "linear2.bias"])
        x = x + drop(h, 4 * l + 3)
    n = f"{prefix}transformer_encoder.norm."
    return F.layer_norm(x, (D,), sd[n + "weight"], sd[n + "bias"], LN_EPS)


def pose_head(x, sd, prefix):
    """models/modules/regressor.py:58-68 (quat, norm_rot=True, noise_dim=0)."""
    h = F.leaky_relu(F.linear(x, sd[prefix + "fc_layers.0.weight"], sd[prefix + "fc_layers.0.bias"]), 0.2)
    h = F.leaky_relu(F.linear(h, sd[prefix + "fc_layers.2.weight"], sd[prefix + "fc_layers.2.bias"]), 0.2)
    rot = F.normalize(F.linear(h, sd[prefix + "rot_head.weight"], sd[prefix + "rot_head.bias"]), p=2, dim=-1)
    return rot, F.linear(h, sd[prefix + "trans_head.weight"], sd[prefix + "trans_head.bias"])


def pn_transformer_forward(sd, batch, num_layers, num_heads, training=True, stats_out=None,
                           encoder="pointnet"):
    """PNTransformer.forward (models/pn_transformer/network.py:59-104), geometric data."""
    pcs, valids = batch["part_pcs"], batch["part_valids"]
    B, P = valids.shape
    mask = valids == 1
    enc = pointnet if encoder == "pointnet" else dgcnn
    feats = enc(pcs[mask], sd, "encoder.", training, stats_out)
    pc_feats = torch.zeros(B, P, feats.shape[-1], dtype=feats.dtype).index_put((mask,), feats)
    corr = transformer_encoder(pc_feats, mask, sd, "corr_module.", num_layers, num_heads)
    rot, trans = pose_head(corr, sd, "pose_predictor.")
    return {"pc_feats": pc_feats, "rot": rot, "trans": trans}


def pn_transformer_loss(sd, batch, num_layers, num_heads, loss_cfg=None, training=True,
                        stats_out=None):
    """forward_pass -> loss_function -> _calc_loss (base_model.py:113-148,240-314,348-387) for the
    geometric configs with sample_iter = 1: returns (scalar loss dict, forward outputs)."""
    out = pn_transformer_forward(sd, batch, num_layers, num_heads, training, stats_out)
    terms = og.calc_loss_geometric(
        og.checked_quat(out["rot"]), out["trans"], batch["part_pcs"],
        og.checked_quat(batch["part_quat"]), batch["part_trans"], batch["part_valids"],
        loss_cfg, training=training)
    return {k: v.mean() for k, v in terms.items()}, out


def adam_step(params, grads, state, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam (weight_decay=0) single-tensor update, the optimiser of base_model.py:406."""
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    bc1, bc2 = 1 - betas[0] ** t, 1 - betas[1] ** t
    with torch.no_grad():
        for k, p in params.items():
            g = grads[k]
            m = state.setdefault("m." + k, torch.zeros_like(p))
            v = state.setdefault("v." + k, torch.zeros_like(p))
            m.lerp_(g, 1 - betas[0])
            v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
            p.addcdiv_(m, (v.sqrt() / math.sqrt(bc2)).add_(eps), value=-lr / bc1)
