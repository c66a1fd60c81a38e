"""Part-relation transformer — mirror of the reference's TransformerEncoder
(multi_part_assembly/models/pn_transformer/transformer.py:4-79): pre-LN encoder layers over <= 20
part tokens with a key-padding mask, final LayerNorm, optional output projection.  Parameters live in
an `nn.TransformerEncoder` so the state_dict keys equal the reference's
(`transformer_encoder.layers.{i}.self_attn.in_proj_weight`, ...); the computation runs on
csrc/transformer.hip (fp32 MFMA GEMMs with LayerNorm / dropout / residual fused in, one-block
attention over the part tokens, deterministic backward).
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn as nn

from . import _lib
from .gradsink import GradSink

_LAYER_PARAMS = ("self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight",
                 "self_attn.out_proj.bias", "linear1.weight", "linear1.bias", "linear2.weight",
                 "linear2.bias", "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias")


class _TransformerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tokens, valid, heads, dropout_p, seed, seed_dev, *params):
        B, P, D = tokens.shape
        L = (len(params) - 2) // len(_LAYER_PARAMS)
        FF = params[4].shape[0]
        dev = tokens.device
        lib = _lib.lib()
        n = ctypes.c_int64()
        _lib.check(lib.mpa_transformer_workspace(B, P, D, heads, FF, L, ctypes.byref(n)),
                   "mpa_transformer_workspace")
        ws = torch.empty(n.value, dtype=torch.float32, device=dev)
        out = torch.empty_like(tokens)
        with torch.cuda.device(dev):
            tok = _lib.KernelTimer.start(f"transformer_forward[{B}x{P}x{D}]")
            st = lib.mpa_transformer_forward(_lib.ptr(tokens), _lib.ptr(valid), _lib.ptr_array(params), B, P, D,
                                             heads, FF, L, float(dropout_p), int(seed),
                                             None if seed_dev is None else _lib.ptr(seed_dev), _lib.ptr(ws),
                                             _lib.ptr(out), _lib.current_stream(dev))
            _lib.KernelTimer.stop(tok)
        _lib.check(st, "mpa_transformer_forward")
        ctx.meta = (heads, FF, L, float(dropout_p), int(seed))
        ctx.params = params  # the Parameter objects themselves (GradSink writes into their .grad)
        GradSink.note_use(params)
        ctx.seed_dev = seed_dev
        ctx.save_for_backward(valid, ws)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        valid, ws = ctx.saved_tensors
        params = ctx.params
        heads, FF, L, dropout_p, seed = ctx.meta
        B, P, D = grad_out.shape
        dev = grad_out.device
        grad_out = grad_out.contiguous()
        grad_tokens = torch.empty_like(grad_out)
        grads, direct = GradSink.outputs(params)
        with torch.cuda.device(dev):
            tok = _lib.KernelTimer.start(f"transformer_backward[{B}x{P}x{D}]")
            st = _lib.lib().mpa_transformer_backward(
                _lib.ptr(grad_out), _lib.ptr(valid), _lib.ptr_array(params), B, P, D, heads, FF, L, dropout_p,
                seed, None if ctx.seed_dev is None else _lib.ptr(ctx.seed_dev), _lib.ptr(ws), _lib.ptr(grad_tokens),
                _lib.ptr_array(grads), _lib.current_stream(dev))
            _lib.KernelTimer.stop(tok)
        _lib.check(st, "mpa_transformer_backward")
        if direct:
            GradSink.delivered(params)
            return (grad_tokens, None, None, None, None, None, *([None] * len(params)))
        return (grad_tokens, None, None, None, None, None, *grads)


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
        self.num_heads, self.norm_first = num_heads, norm_first
        self._calls = 0
        self._seed_dev = None
        # Mixed into the dropout seed so that sibling encoders of one model draw different masks.  The owning model
        # numbers its encoders in module order (`assign_dropout_salts`): a position INSIDE the model, stable from run to
        # run and independent of how many encoders the process built before (a class-level counter was not).
        self._instance = 1
        self._warned = False
        # shapes csrc/transformer.hip is built for (every shipped config: 256 / 8 heads / 1024 / pre-LN)
        self.native = (norm_first and d_model % 64 == 0 and ffn_dim % 64 == 0 and d_model % num_heads == 0
                       and d_model // num_heads <= 64 and num_layers <= 16)

    @staticmethod
    def assign_dropout_salts(model):
        """Number the TransformerEncoder modules of `model` 1, 2, ... in `model.modules()` order."""
        k = 0
        for m in model.modules():
            if isinstance(m, TransformerEncoder):
                k += 1
                m._instance = k

    def _next_seed(self):
        self._calls += 1
        return (torch.initial_seed() * 0x9E3779B1 + self._calls * 0x85EBCA77
                + self._instance * 0xC2B2AE3D27D4EB4F) & 0x7FFFFFFFFFFFFFFF

    def advance_seed(self):
        """New dropout seed for the next replay of a captured step (a counter hashed with torch's seed), written to
        device memory."""
        seed = self._next_seed()
        if self._seed_dev is not None:
            self._seed_dev.fill_(seed)

    def _dropout_p(self):
        """The (single) dropout probability of the stack; 0 in eval mode."""
        if not self.training:
            return 0.0
        ps = set()
        for layer in self.transformer_encoder.layers:
            ps.update((layer.dropout.p, layer.dropout1.p, layer.dropout2.p, layer.self_attn.dropout))
        if len(ps) != 1:
            raise NotImplementedError(f"the HIP transformer uses one dropout probability, found {sorted(ps)}")
        return float(ps.pop())

    def _params(self):
        enc = self.transformer_encoder
        ps = []
        for layer in enc.layers:
            named = dict(layer.named_parameters())
            ps += [named[k] for k in _LAYER_PARAMS]
        return ps + [enc.norm.weight, enc.norm.bias]

    def forward(self, tokens, valid_masks):
        """tokens [B, N, C]; valid_masks [B, N] bool (True = real part), or the float validity matrix itself (a part is
        real iff its entry == 1: the kernels apply the reference's `part_valids == 1` themselves, which spares the
        compare and the cast), or None -> [B, N, C]."""
        if not tokens.is_cuda:
            raise RuntimeError("TransformerEncoder: only CUDA (HIP) tensors are supported — no CPU fallback")
        if valid_masks is not None:
            assert valid_masks.shape == tokens.shape[:2]
        if not self.native or tokens.shape[1] > 64:
            # configurations outside the HIP kernels' instantiation (post-LN, odd widths): library ops — said aloud
            if not self._warned:
                import warnings
                warnings.warn("TransformerEncoder: configuration outside csrc/transformer.hip (needs pre-LN, widths "
                              "multiple of 64, head dim <= 64, <= 64 tokens, <= 16 layers); running on library ops")
                self._warned = True
            pad = None if valid_masks is None else ~(valid_masks if valid_masks.dtype == torch.bool else valid_masks == 1)
            return self.out_fc(self.transformer_encoder(tokens, src_key_padding_mask=pad))
        B, P, _ = tokens.shape
        valid = (torch.ones(B * P, device=tokens.device) if valid_masks is None
                 else valid_masks.reshape(-1).float())
        p = self._dropout_p()
        seed, seed_dev = 0, None
        if p > 0.0 and not torch.cuda.is_current_stream_capturing():
            # eager call: the seed travels BY VALUE with the launches (the backward regenerates the masks from the value
            # its own forward saved, so a second forward before that backward cannot disturb it) — no device write
            seed = self._next_seed()
            self._salt = 0
            if self._seed_dev is None or self._seed_dev.device != tokens.device:  # (once: ready for a later capture)
                self._seed_dev = torch.zeros(1, dtype=torch.int64, device=tokens.device)
        elif p > 0.0:
            # capture: the seed lives in device memory so that the captured step draws fresh masks on every replay (the
            # Trainer calls advance_seed() between replays); the k-th call of the module inside one step reads
            # seed + k * odd constant at replay time
            if self._seed_dev is None or self._seed_dev.device != tokens.device:
                raise RuntimeError("TransformerEncoder: run one eager training forward on this device before capturing "
                                   "a step (the dropout seed's device word is allocated there, outside the graph)")
            seed_dev = self._seed_dev + (getattr(self, "_salt", 0) * 0x632BE59BD9B4E019 & 0x3FFFFFFFFFFFFFFF)
            self._salt = getattr(self, "_salt", 0) + 1
        out = _TransformerFn.apply(tokens.float().contiguous(), valid.contiguous(), self.num_heads, p, seed, seed_dev,
                                   *self._params())
        return self.out_fc(out)
