"""Short-sequence self-attention on MI355X (second step).

``attention(q, k, v, scale)`` = ``softmax(scale * q k^T) v`` for ``[B, H, L, 64]`` bf16 operands with ``L <= 256``,
no mask and no dropout -- the shape of every attention call in the encoders of the headline workload (ViT-B/16
L = 197, mmlearn's own ViT L = 196 / 169, BERT L = 77).  One workgroup keeps a whole (batch, head) problem on chip
(``csrc/attention.hip``).  Registered with HF transformers' ``AttentionInterface`` as ``"mmlearn_hip"`` so that
``CLIPVisionModel`` & co. can select it per config; calls it cannot serve (mask, dropout, other head dims or dtypes)
are forwarded to SDPA unchanged.
"""

from __future__ import annotations

from typing import Optional

import torch

from . import kernels as K


def supported(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, attention_mask=None, dropout: float = 0.0, is_causal=False) -> bool:
    return (q.is_cuda and q.dtype == torch.bfloat16 and k.dtype == torch.bfloat16 and v.dtype == torch.bfloat16 and q.dim() == 4
            and q.shape == k.shape == v.shape and q.shape[-1] == 64 and q.shape[-2] <= 256 and attention_mask is None
            and not dropout and not is_causal and q.stride(-1) == 1 and k.stride(-1) == 1 and v.stride(-1) == 1
            and all(s % 8 == 0 for t in (q, k, v) for s in t.stride()[:3]))


class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, scale):
        out, lse = K.attn_fwd(q, k, v, scale)   # out: [B, L, H, 64] contiguous
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        dq, dk, dv = K.attn_bwd(q, k, v, out, lse, dout.contiguous(), ctx.scale)
        return dq, dk, dv, None


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: Optional[float] = None) -> torch.Tensor:
    """q, k, v: ``[B, H, L, 64]`` bf16 (any strides with a contiguous last dim).  Returns ``[B, L, H, 64]`` contiguous,
    i.e. already in the layout the output projection wants (``.reshape(B, L, H*64)`` is free)."""
    K.require_gpu(q)
    if not supported(q, k, v):
        raise ValueError("mmlearn_amd.attention: need bf16 [B, H, L<=256, 64] operands with contiguous last dim")
    return _Attention.apply(q, k, v, float(scale if scale is not None else q.shape[-1] ** -0.5))


def hf_attention_forward(module, query, key, value, attention_mask, dropout: float = 0.0, scaling: Optional[float] = None,
                         is_causal: Optional[bool] = None, **kwargs):
    """transformers ``AttentionInterface`` entry: same contract as ``sdpa_attention_forward`` (returns
    ``(attn_output [B, L, H, dh], None)``).  Unsupported calls fall through to SDPA."""
    if supported(query, key, value, attention_mask, dropout if module.training else 0.0, bool(is_causal)):
        return attention(query, key, value, scaling), None
    from transformers.integrations.sdpa_attention import sdpa_attention_forward

    return sdpa_attention_forward(module, query, key, value, attention_mask, dropout=dropout, scaling=scaling, is_causal=is_causal, **kwargs)


def register_hf_attention(name: str = "mmlearn_hip") -> str:
    """Make ``config._attn_implementation = "mmlearn_hip"`` available to HF models."""
    from transformers import AttentionInterface

    AttentionInterface.register(name, hf_attention_forward)
    return name
