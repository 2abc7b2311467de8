"""ORACLE (test infrastructure, NOT product code): numpy restatement of the
reference's I-JEPA path ops.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  Pinned against ``tests/golden/g6_ijepa.npz``,
``g7_masks.npz`` and ``g8_ema.npz`` (outputs of the reference itself).
"""

from __future__ import annotations

import math

import numpy as np


# mmlearn/datasets/processors/masking.py:241-287
def apply_masks(x: np.ndarray, masks):
    """x [B,N,D]; masks: list of (N,), (1,N) or (B,N) 0/1 arrays -> [len(masks)*B, keep, D]."""
    B = x.shape[0]
    out = []
    for m in masks:
        m = np.asarray(m)
        if m.ndim == 1:
            m = m[None]
        if m.shape[0] == 1 and B > 1:
            m = np.broadcast_to(m, (B, m.shape[1]))
        m = m.astype(bool)
        out.append(x[m].reshape(B, -1, x.shape[-1]))
    return np.concatenate(out, 0)


def masks_to_indices(mask: np.ndarray):
    """Row-wise sorted keep indices of a (B,N) 0/1 mask with equal keep counts."""
    mask = np.asarray(mask).astype(bool)
    if mask.ndim == 1:
        mask = mask[None]
    cnt = mask.sum(1)
    assert (cnt == cnt[0]).all()
    return np.stack([np.nonzero(r)[0] for r in mask]).astype(np.int32)


# mmlearn/datasets/processors/transforms.py:55-79
def repeat_interleave_batch(x: np.ndarray, b: int, repeat: int):
    n = len(x) // b
    return np.concatenate([np.concatenate([x[i * b:(i + 1) * b]] * repeat, 0) for i in range(n)], 0)


# F.layer_norm(h, h.size()[-1:]) -- tasks/ijepa.py:234 (no affine, eps 1e-5, biased variance)
def layer_norm(x: np.ndarray, eps: float = 1e-5):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps)


# F.smooth_l1_loss(z, t) (beta=1, mean) -- tasks/ijepa.py:86,256
def smooth_l1(z: np.ndarray, t: np.ndarray):
    d = z - t
    ad = np.abs(d)
    per = np.where(ad < 1.0, 0.5 * d * d, ad - 0.5)
    loss = per.mean()
    dz = np.where(ad < 1.0, d, np.sign(d)) / d.size
    return float(loss), dz


def mse(z: np.ndarray, t: np.ndarray):
    d = z - t
    return float((d * d).mean()), 2.0 * d / d.size


# tasks/ijepa.py:232-238
def ijepa_target(h: np.ndarray, pred_masks, n_enc_masks: int = 1):
    B = h.shape[0]
    return repeat_interleave_batch(apply_masks(layer_norm(h), pred_masks), B, repeat=n_enc_masks)


# modules/encoders/vision.py:545-560 (the sequence the predictor blocks consume)
def predictor_assemble(x_embed: np.ndarray, pos_embed: np.ndarray, mask_token: np.ndarray, enc_masks, pred_masks):
    """x_embed [len(enc)*B, n_ctxt, Dp] (after predictor_embed); pos_embed [1,N,Dp]; mask_token [1,1,Dp]."""
    b = len(x_embed) // len(enc_masks)
    pos = np.repeat(pos_embed, b, axis=0)
    x = x_embed + apply_masks(pos, enc_masks)
    pe = repeat_interleave_batch(apply_masks(pos, pred_masks), b, repeat=len(enc_masks))
    pred_tokens = mask_token + pe
    x = np.concatenate([x] * len(pred_masks), 0)
    return np.concatenate([x, pred_tokens], 1)


def predictor_assemble_bwd(d_seq: np.ndarray, n_rows_x: int, n_ctxt: int, n_pred_masks: int):
    """Gradients of predictor_assemble w.r.t. x_embed and mask_token."""
    d_x = d_seq[:, :n_ctxt].reshape(n_pred_masks, n_rows_x, n_ctxt, -1).sum(0)
    d_tok = d_seq[:, n_ctxt:].sum((0, 1))[None, None]
    return d_x, d_tok


# ----------------------------------------------------------------------------
# mmlearn/datasets/processors/masking.py:290-415.  The RNG call sequence is the
# contract (G7): one randint on the GLOBAL generator for the seed, two rand on a
# private generator (pred size, enc size), then per block two randint on the
# GLOBAL generator (top, left).  torch is used ONLY as the RNG.
def ijepa_masks(batch_size: int = 1, input_size=(224, 224), patch_size: int = 16, enc_mask_scale=(0.85, 1.0),
                pred_mask_scale=(0.15, 0.2), aspect_ratio=(0.75, 1.5), nenc: int = 1, npred: int = 4):
    import torch

    H, W = input_size[0] // patch_size, input_size[1] // patch_size
    seed = torch.randint(0, 2**32, (1,)).item()
    g = torch.Generator().manual_seed(seed)

    def block_size(scale, ar):
        r = torch.rand(1, generator=g).item()
        max_keep = int(H * W * (scale[0] + r * (scale[1] - scale[0])))
        a = ar[0] + r * (ar[1] - ar[0])
        h = int(round(math.sqrt(max_keep * a)))
        w = int(round(math.sqrt(max_keep / a)))
        return min(h, H - 1), min(w, W - 1)

    def block_mask(hw):
        h, w = hw
        top = torch.randint(0, H - h, (1,)).item()
        left = torch.randint(0, W - w, (1,)).item()
        m = np.zeros((H, W), np.int32)
        m[top:top + h, left:left + w] = 1
        return np.broadcast_to(m.reshape(1, -1), (batch_size, H * W)).copy()

    p_size = block_size(pred_mask_scale, aspect_ratio)
    e_size = block_size(enc_mask_scale, (1.0, 1.0))
    pred = [block_mask(p_size) for _ in range(npred)]
    enc = [block_mask(e_size) for _ in range(nenc)]
    return {"encoder_masks": enc, "predictor_masks": pred}


# ----------------------------------------------------------------------------
# mmlearn/modules/ema.py:79-89,132-177
def annealed_rate(start: float, end: float, step: int, total: int):
    return end - (end - start) * (1 - step / total)


class EmaOracle:
    """State machine of ExponentialMovingAverage on dicts of numpy arrays.

    ``true_ema=False`` reproduces quirk Q1: the reference
    tests ``param.requires_grad`` on ``state_dict()`` tensors, which is always
    False, so every tensor takes the copy branch (ema.py:147-148).
    """

    def __init__(self, state: dict, ema_decay: float, ema_end_decay: float, anneal_end_step: int, true_ema: bool = False,
                 trainable=None):
        self.state = {k: np.array(v, copy=True) for k, v in state.items()}
        self.decay = self.ema_decay = ema_decay
        self.ema_end_decay, self.anneal_end_step = ema_end_decay, anneal_end_step
        self.num_updates, self.true_ema = 0, true_ema
        self.trainable = set(trainable or [])

    def step(self, student: dict):
        if self.decay < 1:
            for k, p in student.items():
                e = self.state[k]
                if self.true_ema and k in self.trainable and np.issubdtype(e.dtype, np.floating):
                    self.state[k] = (e.astype(np.float32) * np.float32(self.decay)
                                     + p.astype(np.float32) * np.float32(1 - self.decay)).astype(e.dtype)
                else:
                    self.state[k] = p.astype(e.dtype).copy()
            self.num_updates += 1
        if self.ema_decay != self.ema_end_decay:
            if self.num_updates >= self.anneal_end_step:
                self.decay = self.ema_end_decay
            else:
                self.decay = annealed_rate(self.ema_decay, self.ema_end_decay, self.num_updates, self.anneal_end_step)
