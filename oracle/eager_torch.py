"""ORACLE (test infrastructure, NOT product code): op-by-op torch restatement of the reference's
contrastive loss (the ATen sequence K1-K7 of SURVEY.md 2.3), used

* as the CPU baseline of ``bench.py`` (``cpu_baseline.kind = "port"``): the same op sequence the
  reference runs, on the host cores;
* by GPU tests as the "reference PyTorch-ROCm eager path" to time the HIP path against.

Only ``tests/`` , ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` import this.
Checked against the golden vectors in ``tests/test_oracle_golden.py`` (single process and, through
``contrastive_loss_dist_torch``, the world_size>1 gather path).
"""

from __future__ import annotations

import torch
import torch.distributed as dist
import torch.nn.functional as F


# mmlearn/datasets/core/example.py:160-166
def find_matching_indices(a: torch.Tensor, b: torch.Tensor):
    matches = torch.all(a.unsqueeze(1) == b.unsqueeze(0), dim=-1)
    return torch.where(matches)


# torchmetrics 1.6.2 _safe_matmul (contract restated; see oracle/clip_oracle.py header)
def _safe_matmul(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    if x.dtype == torch.float16 or y.dtype == torch.float16:
        return (x.float() @ y.T.float()).half()
    return x @ y.T


# mmlearn/modules/losses/contrastive.py:59-160, world_size == 1
def contrastive_loss_torch(embeddings: dict, example_ids: dict, logit_scale: torch.Tensor, pairs, l2_normalize: bool = False):
    """embeddings keyed by modality name; pairs: list of ((a, b), weight)."""
    if l2_normalize:
        embeddings = {k: F.normalize(v, p=2, dim=-1) for k, v in embeddings.items()}
    losses = []
    for (ma, mb), w in pairs:
        if ma not in embeddings or mb not in embeddings:
            continue
        ia, ib = find_matching_indices(example_ids[ma], example_ids[mb])
        if ia.numel() == 0:
            continue
        fa, fb = embeddings[ma][ia], embeddings[mb][ib]
        logits_a = logit_scale * _safe_matmul(fa, fb)
        logits_b = logit_scale * _safe_matmul(fb, fa)
        labels = torch.arange(logits_a.shape[-1], device=logits_a.device, dtype=torch.long)
        losses.append((F.cross_entropy(logits_a, labels) + F.cross_entropy(logits_b, labels)) / 2 * w)
    if not losses:
        return torch.tensor(0.0, device=logit_scale.device, dtype=next(iter(embeddings.values())).dtype)
    return torch.stack(losses).sum()


# the same with the reference's default distributed cell (local_loss=False, gather_with_grad=False):
# every rank gathers all embeddings (own shard re-inserted to keep grad) and builds the full [N, N] logits
def contrastive_loss_dist_torch(embeddings: dict, example_ids: dict, logit_scale: torch.Tensor, pairs):
    world, rank = dist.get_world_size(), dist.get_rank()

    def gather(t):
        out = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(out, t.contiguous())
        out[rank] = t
        return torch.cat(out, 0)

    all_e = {k: gather(v) for k, v in embeddings.items()}
    all_i = {k: gather(v) for k, v in example_ids.items()}
    losses = []
    for (ma, mb), w in pairs:
        ia, ib = find_matching_indices(all_i[ma], all_i[mb])
        fa, fb = all_e[ma][ia], all_e[mb][ib]
        logits_a = logit_scale * _safe_matmul(fa, fb)
        logits_b = logits_a.T
        labels = torch.arange(logits_a.shape[-1], device=logits_a.device, dtype=torch.long)
        losses.append((F.cross_entropy(logits_a, labels) + F.cross_entropy(logits_b, labels)) / 2 * w)
    return torch.stack(losses).sum()


class EagerContrastiveLoss(torch.nn.Module):
    """Module form with the reference's call signature (embeddings keyed by '<mod>_embedding')."""

    def __init__(self, l2_normalize: bool = False):
        super().__init__()
        self.l2_normalize = l2_normalize

    def forward(self, embeddings, example_ids, logit_scale, modality_loss_pairs):
        emb = {k[: -len("_embedding")]: v for k, v in embeddings.items()}
        pairs = [(tuple(p.modalities), float(p.weight)) for p in modality_loss_pairs]
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return contrastive_loss_dist_torch(emb, example_ids, logit_scale, pairs)
        return contrastive_loss_torch(emb, example_ids, logit_scale, pairs, self.l2_normalize)
