"""Short-sequence self-attention on MI355X (second step).

``attention(q, k, v, scale, dropout_p)`` = ``dropout(softmax(scale * q k^T)) v`` for ``[B, H, L, 64]`` bf16 operands with
``L <= 256`` and no mask -- the shape of every attention call in the encoders of the headline workload (ViT-B/16
L = 197, mmlearn's own ViT L = 196 / 169, BERT L = 77 with attention dropout 0.1).  One workgroup keeps a whole
(batch, head) problem on chip (``csrc/attention.hip``); the dropout mask is a counter-based function of a per-call seed
drawn from torch's CPU generator (so ``torch.manual_seed`` reproduces it), regenerated in the backward.  Registered with
HF transformers' ``AttentionInterface`` as ``"mmlearn_hip"`` so that ``CLIPVisionModel`` / ``BertModel`` & co. can select
it per config; calls it cannot serve (attention mask, causal, other head dims or dtypes) are forwarded to SDPA unchanged.
"""

from __future__ import annotations

from typing import Optional

import torch

from . import kernels as K


def supported(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, attention_mask=None, dropout: float = 0.0, is_causal=False) -> bool:
    return (q.is_cuda and q.dtype == torch.bfloat16 and k.dtype == torch.bfloat16 and v.dtype == torch.bfloat16 and q.dim() == 4
            and q.shape == k.shape == v.shape and q.shape[-1] == 64 and q.shape[-2] <= 256 and attention_mask is None
            and 0.0 <= float(dropout) < 1.0 and not is_causal and q.stride(-1) == 1 and k.stride(-1) == 1 and v.stride(-1) == 1
            and all(s % 8 == 0 for t in (q, k, v) for s in t.stride()[:3]))


class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, scale, dropout_p, seed):
        out, lse = K.attn_fwd(q, k, v, scale, dropout_p, seed)   # out: [B, L, H, 64] contiguous
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.scale, ctx.dropout_p, ctx.seed = scale, dropout_p, seed
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        dq, dk, dv = K.attn_bwd(q, k, v, out, lse, dout.contiguous(), ctx.scale, ctx.dropout_p, ctx.seed)
        return dq, dk, dv, None, None, None


class _AttentionPacked(torch.autograd.Function):
    """Same kernels on a packed ``[B, L, 3, H, 64]`` QKV tensor (the output of ONE fused projection); the backward writes
    dq / dk / dv straight into one packed gradient, so autograd sees a single tensor in and out."""

    @staticmethod
    def forward(ctx, qkv, scale, dropout_p, seed):
        q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))   # [B, H, L, 64] views, row stride 3*H*64
        out, lse = K.attn_fwd(q, k, v, scale, dropout_p, seed)
        ctx.save_for_backward(qkv, out, lse)
        ctx.scale, ctx.dropout_p, ctx.seed = scale, dropout_p, seed
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))
        return K.attn_bwd(q, k, v, out, lse, dout.contiguous(), ctx.scale, ctx.dropout_p, ctx.seed, packed=True), None, None, None


def attention_qkvpacked(qkv: torch.Tensor, scale: Optional[float] = None, dropout_p: float = 0.0, seed: Optional[int] = None) -> torch.Tensor:
    """``qkv``: ``[B, L, 3, H, 64]`` bf16 contiguous (``Linear(E, 3E)(x).view(B, L, 3, H, 64)``) -> ``[B, L, H, 64]``."""
    K.require_gpu(qkv)
    if not (qkv.dim() == 5 and qkv.shape[2] == 3 and qkv.shape[-1] == 64 and qkv.shape[1] <= 256 and qkv.dtype == torch.bfloat16
            and qkv.is_contiguous() and 0.0 <= dropout_p < 1.0):
        raise ValueError("mmlearn_amd.attention_qkvpacked: need a contiguous bf16 [B, L<=256, 3, H, 64] tensor")
    if dropout_p > 0.0 and seed is None:
        seed = draw_seed()
    return _AttentionPacked.apply(qkv, float(scale if scale is not None else 0.125), float(dropout_p), int(seed or 0))


def draw_seed() -> int:
    """A 63-bit seed from torch's default CPU generator (no device sync; reproducible under ``torch.manual_seed``)."""
    return int(torch.empty((), dtype=torch.int64).random_().item())


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: Optional[float] = None, dropout_p: float = 0.0,
              seed: Optional[int] = None) -> torch.Tensor:
    """q, k, v: ``[B, H, L, 64]`` bf16 (any strides with a contiguous last dim).  Returns ``[B, L, H, 64]`` contiguous,
    i.e. already in the layout the output projection wants (``.reshape(B, L, H*64)`` is free).  ``dropout_p`` drops
    attention probabilities (training-time semantics of ``F.scaled_dot_product_attention(dropout_p=...)``)."""
    K.require_gpu(q)
    if not supported(q, k, v, dropout=dropout_p):
        raise ValueError("mmlearn_amd.attention: need bf16 [B, H, L<=256, 64] operands with contiguous last dim")
    if dropout_p > 0.0 and seed is None:
        seed = draw_seed()
    return _Attention.apply(q, k, v, float(scale if scale is not None else q.shape[-1] ** -0.5), float(dropout_p), int(seed or 0))


def hf_attention_forward(module, query, key, value, attention_mask, dropout: float = 0.0, scaling: Optional[float] = None,
                         is_causal: Optional[bool] = None, **kwargs):
    """transformers ``AttentionInterface`` entry: same contract as ``sdpa_attention_forward`` (returns
    ``(attn_output [B, L, H, dh], None)``).  Unsupported calls fall through to SDPA."""
    if supported(query, key, value, attention_mask, dropout, bool(is_causal)):
        return attention(query, key, value, scaling, float(dropout)), None
    from transformers.integrations.sdpa_attention import sdpa_attention_forward

    return sdpa_attention_forward(module, query, key, value, attention_mask, dropout=dropout, scaling=scaling, is_causal=is_causal, **kwargs)


def register_hf_attention(name: str = "mmlearn_hip") -> str:
    """Make ``config._attn_implementation = "mmlearn_hip"`` available to HF models."""
    from transformers import AttentionInterface

    AttentionInterface.register(name, hf_attention_forward)
    return name
