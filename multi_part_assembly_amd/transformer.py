"""Part-relation transformer — mirror of the reference's TransformerEncoder
(multi_part_assembly/models/pn_transformer/transformer.py:4-79): pre-LN encoder layers over <= 20
part tokens with a key-padding mask, final LayerNorm, optional output projection.  Parameters live in
an `nn.TransformerEncoder` so the state_dict keys equal the reference's
(`transformer_encoder.layers.{i}.self_attn.in_proj_weight`, ...).
"""
from __future__ import annotations

import torch.nn as nn


class TransformerEncoder(nn.Module):
    def __init__(self, d_model, num_heads, ffn_dim, num_layers, norm_first=True, dropout=0.1,
                 out_dim=None):
        super().__init__()
        layer = nn.TransformerEncoderLayer(d_model=d_model, nhead=num_heads, dim_feedforward=ffn_dim,
                                           dropout=dropout, norm_first=norm_first, batch_first=True)
        self.transformer_encoder = nn.TransformerEncoder(
            layer, num_layers=num_layers, norm=nn.LayerNorm(d_model) if norm_first else None,
            enable_nested_tensor=False)
        self.out_fc = nn.Linear(d_model, out_dim) if out_dim is not None else nn.Identity()

    def forward(self, tokens, valid_masks):
        """tokens [B, N, C]; valid_masks [B, N] bool (True = real part) or None -> [B, N, C]."""
        pad = None
        if valid_masks is not None:
            assert valid_masks.shape == tokens.shape[:2]
            pad = ~valid_masks
        return self.out_fc(self.transformer_encoder(tokens, src_key_padding_mask=pad))
