"""CPU oracle for the short-sequence attention kernels (TEST INFRASTRUCTURE ONLY -- never imported by the product).

The op is the one the reference's encoders run per transformer block: ``softmax(q k^T * scale)`` (+ dropout on the
probabilities) ``@ v`` (mmlearn/modules/layers/attention.py:60-75 for mmlearn's own ViT; HF CLIP / BERT through
``F.scaled_dot_product_attention``).  The float part is restated in plain fp32/fp64 torch; the dropout keep-mask of
``mmlearn_amd/csrc/attention.hip`` (a counter-based hash, so there is no reference stream to match) is restated in
numpy integer arithmetic bit for bit, so the tests can compare a dropped-out forward / backward exactly.
"""

from __future__ import annotations

import numpy as np
import torch

_M32 = np.uint64(0xFFFFFFFF)
_M24 = np.uint64(0xFFFFFF)


def _fmix32(h):
    h = np.asarray(h, dtype=np.uint64) & _M32
    h = h ^ (h >> np.uint64(16))
    h = (h * np.uint64(0x85EBCA6B)) & _M32
    h = h ^ (h >> np.uint64(13))
    h = (h * np.uint64(0xC2B2AE35)) & _M32
    return h ^ (h >> np.uint64(16))


def drop_key(seed: int, bh):
    lo, hi = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    return _fmix32(lo ^ ((np.asarray(bh, dtype=np.uint64) * np.uint64(0x9E3779B1)) & _M32)) ^ hi


def drop_word(key, i, jp):
    idx = np.asarray(i, dtype=np.uint64) * np.uint64(128) + np.asarray(jp, dtype=np.uint64)
    a = ((idx & _M24) * np.uint64(0x9E3779) + key) & _M32
    a = a ^ (a >> np.uint64(16))
    a = ((a & _M24) * np.uint64(0xB5297A) + np.uint64(0x1B873593)) & _M32
    a = a ^ (a >> np.uint64(15))
    a = ((a & _M24) * np.uint64(0x68E31D)) & _M32
    return a ^ (a >> np.uint64(16))


def drop_threshold(p: float) -> int:
    return int(np.rint(np.float32(p) * np.float32(65536.0))) if p > 0 else 0


def keep_mask(seed: int, B: int, H: int, L: int, p: float) -> np.ndarray:
    """bool [B, H, L, L]: True where attention probability (i, j) survives dropout."""
    thr = np.uint64(drop_threshold(p))
    i, jp = np.meshgrid(np.arange(L), np.arange((L + 1) // 2), indexing="ij")
    out = np.empty((B * H, L, L), dtype=bool)
    for bh in range(B * H):
        w = drop_word(drop_key(seed, bh), i, jp)
        lo = (w & np.uint64(0xFFFF)) >= thr
        hi = (w >> np.uint64(16)) >= thr
        full = np.empty((L, 2 * ((L + 1) // 2)), dtype=bool)
        full[:, 0::2], full[:, 1::2] = lo, hi
        out[bh] = full[:, :L]
    return out.reshape(B, H, L, L)


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float, dropout_p: float = 0.0, seed: int = 0,
              dtype=torch.float32, key_mask: torch.Tensor | None = None, causal: bool = False) -> torch.Tensor:
    """[B, H, L, dh] operands -> [B, L, H, dh]; differentiable (autograd supplies the backward oracle).

    ``key_mask``: bool ``[B, L]``, True = the key is attended -- the tokenizer's padding mask the reference's text towers forward
    (mmlearn/modules/encoders/text.py:160-165, clip.py:104-107, 329-346), i.e. ``attn_mask=key_mask[:, None, None, :]`` of
    ``F.scaled_dot_product_attention`` / HF's ``[B, 1, 1, L]`` extended mask.  Masked logits are set to ``finfo.min`` of the compute
    type, not -inf (HF's additive convention): a sample whose keys are ALL masked averages V uniformly instead of producing NaN.
    ``causal``: key j > query i is masked the same way (HF CLIP's text tower)."""
    s = (q.to(dtype) @ k.to(dtype).transpose(-1, -2)) * scale
    if key_mask is not None or causal:
        B, H, L, _ = q.shape
        allowed = torch.ones(B, 1, L, L, dtype=torch.bool, device=s.device)
        if key_mask is not None:
            allowed = allowed & key_mask.to(s.device).bool()[:, None, None, :]
        if causal:
            allowed = allowed & torch.ones(L, L, dtype=torch.bool, device=s.device).tril()
        s = s.masked_fill(~allowed, torch.finfo(dtype).min)
    p = torch.softmax(s, dim=-1)
    if dropout_p > 0:
        B, H, L, _ = q.shape
        thr = drop_threshold(dropout_p)
        keep = torch.from_numpy(keep_mask(seed, B, H, L, dropout_p)).to(p.device)
        p = p * keep.to(dtype) * (65536.0 / (65536.0 - thr))
    return (p @ v.to(dtype)).transpose(1, 2)


def hidden_keep_mask(seed: int, rows: int, d: int, p: float) -> np.ndarray:
    """bool [rows, d]: the keep-mask of the dropout fused into add + LayerNorm (csrc/encoder_ops.hip): key = (seed, row),
    one word per column pair."""
    thr = np.uint64(drop_threshold(p))
    jp = np.arange((d + 1) // 2)
    out = np.empty((rows, 2 * ((d + 1) // 2)), dtype=bool)
    keys = drop_key(seed, np.arange(rows))
    for r in range(rows):
        w = drop_word(keys[r], 0, jp)
        out[r, 0::2] = (w & np.uint64(0xFFFF)) >= thr
        out[r, 1::2] = (w >> np.uint64(16)) >= thr
    return out[:, :d]
