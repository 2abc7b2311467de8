"""Tiny stand-in encoders shared by the golden generator and the tests.

Own code (not from the reference).  The reference tasks accept any module that
honours the encoder contract (``forward(inputs: dict) -> list-like`` whose
``[0]`` is the embedding, docs/user_guide.md:120-125), so the SAME tiny models
can be driven by the reference (in ``gen_golden.py``) and by this repo's tasks
(in the tests); only their weights and the reference's outputs are stored.
"""

from __future__ import annotations

import torch
from torch import nn


class FlatMLPEncoder(nn.Module):
    """images [B,3,H,W] -> flatten -> 2-layer MLP."""

    def __init__(self, key: str, in_dim: int, hidden: int, out: int):
        super().__init__()
        self.key = key
        self.net = nn.Sequential(nn.Linear(in_dim, hidden), nn.GELU(), nn.Linear(hidden, out))

    def forward(self, inputs):
        return (self.net(inputs[self.key].float().flatten(1)),)


class TokenMLPEncoder(nn.Module):
    """token ids [B,L] -> embedding mean -> 2-layer MLP."""

    def __init__(self, key: str, vocab: int, dim: int, out: int):
        super().__init__()
        self.key = key
        self.emb = nn.Embedding(vocab, dim)
        self.mlp = nn.Sequential(nn.Linear(dim, dim), nn.GELU(), nn.Linear(dim, out))

    def forward(self, inputs):
        return (self.mlp(self.emb(inputs[self.key]).mean(1)),)


class _PatchEmbed(nn.Module):
    def __init__(self, img: int, patch: int, dim: int):
        super().__init__()
        self.patch = patch
        self.num_patches = (img // patch) ** 2
        self.proj = nn.Linear(3 * patch * patch, dim)

    def forward(self, x):
        p = self.patch
        B, C, H, W = x.shape
        x = x.unfold(2, p, p).unfold(3, p, p)  # B,C,H/p,W/p,p,p
        x = x.permute(0, 2, 3, 1, 4, 5).reshape(B, -1, C * p * p)
        return self.proj(x)


class TinyPatchEncoder(nn.Module):
    """Patchify -> +pos -> (optional) keep-mask -> token MLP -> LayerNorm.

    ``mask_fn(x, masks)`` is the patch-gather used on the context branch
    (reference: ``apply_masks``; here: this repo's gather op).
    """

    def __init__(self, embed_dim: int = 32, img: int = 224, patch: int = 16, num_heads: int = 2, mask_fn=None):
        super().__init__()
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.patch_embed = _PatchEmbed(img, patch, embed_dim)
        self.pos_embed = nn.Parameter(0.02 * torch.randn(1, self.patch_embed.num_patches, embed_dim), requires_grad=False)
        self.mlp = nn.Sequential(nn.Linear(embed_dim, 2 * embed_dim), nn.GELU(), nn.Linear(2 * embed_dim, embed_dim))
        self.norm = nn.LayerNorm(embed_dim)
        self.mask_fn = mask_fn

    def forward(self, inputs):
        x = self.patch_embed(inputs["rgb"]) + self.pos_embed
        masks = inputs.get("rgb_mask")
        if masks is not None:
            if not isinstance(masks, list):
                masks = [masks]
            x = self.mask_fn(x, masks)
        x = x + self.mlp(x)
        return (self.norm(x), None)


class SimplePredictor(nn.Module):
    """A block-less predictor with the reference predictor's parameter names
    (predictor_embed / mask_token / predictor_pos_embed / predictor_norm /
    predictor_proj), to be wrapped by this repo's predictor front-end."""

    def __init__(self, num_patches: int, embed_dim: int, predictor_embed_dim: int):
        super().__init__()
        self.num_patches, self.embed_dim, self.num_heads = num_patches, embed_dim, 1
        self.predictor_embed = nn.Linear(embed_dim, predictor_embed_dim)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, predictor_embed_dim))
        self.predictor_pos_embed = nn.Parameter(torch.zeros(1, num_patches, predictor_embed_dim), requires_grad=False)
        self.predictor_blocks = nn.ModuleList([])
        self.predictor_norm = nn.LayerNorm(predictor_embed_dim)
        self.predictor_proj = nn.Linear(predictor_embed_dim, embed_dim)
