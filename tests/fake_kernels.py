"""TEST DOUBLE of ``mmlearn_amd.kernels`` on CPU tensors (float64), used ONLY by the multi-process gloo
tests to exercise the host-side sharding / exchange logic of ``mmlearn_amd.losses`` without a GPU.

It restates the *contract* of each kernel entry point (include/mmlearn_hip.h) with plain torch ops; it
is installed by monkeypatching ``mmlearn_amd.losses.K`` from the tests and is never importable from
the product.  The HIP kernels themselves are checked against the oracle in the ``-m gpu`` tests.
"""

from __future__ import annotations

from typing import Optional, Sequence

import torch

from mmlearn_amd.kernels import Direction, Match  # dataclasses only  # noqa: F401
from oracle import clip_oracle as co

MAX_DIRS_PER_CALL = 8
CALLS = {"pack_rows": 0, "clip_forward": 0, "clip_backward": 0, "match_ids": 0, "transposes": 0}
ONE_KERNEL_PAIRS = True
ON_CHIP_BACKWARD = False   # what backward_recomputes_on_chip answers: the tests flip it to walk both of the host's packing branches


def require_gpu(t, what="tensor"):
    return None


def round_up(a, b):
    return (a + b - 1) // b * b


def match_ids(ids_a: torch.Tensor, ids_b: torch.Tensor) -> Match:
    CALLS["match_ids"] += 1
    n_a, n_b = ids_a.shape[0], ids_b.shape[0]
    if n_a == 0 or n_b == 0:
        return Match(0, False, torch.empty(0, dtype=torch.int32), torch.empty(0, dtype=torch.int32))
    ia, ib = co.find_matching_indices(ids_a.numpy(), ids_b.numpy())
    n = len(ia)
    ident = n == n_a == n_b and (ia == range(n)).all() and (ib == range(n)).all()
    if ident:
        return Match(n, True, None, None)
    rep_a = len(set(ia.tolist())) < n
    rep_b = len(set(ib.tolist())) < n
    return Match(n, False, torch.from_numpy(ia).int(), torch.from_numpy(ib).int(), rep_a, rep_b)


def pack_rows(src, idx, r, normalize, compute, want_transpose):
    CALLS["pack_rows"] += 1
    n_src, d = src.shape
    rows = src.detach().double()
    rows = rows[idx[:r].long()] if idx is not None else rows[:r]
    if normalize:
        rows = rows / rows.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    r_pad, k_pad = round_up(max(r, 1), 128), round_up(d, 64)
    dst = torch.zeros(r_pad, k_pad, dtype=torch.float64)
    dst[:r, :d] = rows
    CALLS["transposes"] += int(bool(want_transpose))
    return dst, (dst.T.contiguous() if want_transpose else None)


def pack_rows_many(reqs, compute):
    return [pack_rows(*q[:4], compute, q[4]) for q in reqs]


def pair_runs_untied(n, d, compute, n_dirs=2):
    return False


def backward_recomputes_on_chip(r, c, d, compute, n_dirs=2):
    """contract of kernels.backward_recomputes_on_chip: True = the directions need no transposed operand (y_t may be None)"""
    return ON_CHIP_BACKWARD


def clip_forward(dirs: Sequence[Direction], d: int, compute: int, scale: torch.Tensor, loss_weights=None) -> None:
    CALLS["clip_forward"] += 1
    s = scale.double().item()
    for dr in dirs:
        t = dr.x[: dr.r] @ dr.y[: dr.c].T
        v = s * t
        if dr.mode == 1:
            per = _align_rows(dr, v)[0]
            nb = (dr.r + 255) // 256
            dr.loss_part = torch.stack([per[k * 256:(k + 1) * 256].sum() for k in range(nb)]).float()
            continue
        lse = torch.logsumexp(v, dim=1)
        lab = dr.label_off + torch.arange(dr.r)
        dr.lse = lse.float()
        dr.diag = v[torch.arange(dr.r), lab].float()
        per = lse - v[torch.arange(dr.r), lab]
        nb = (dr.r + 255) // 256
        dr.loss_part = torch.stack([per[k * 256:(k + 1) * 256].sum() for k in range(nb)]).float()


def _align_rows(dr, v):
    """per-row alignment loss and the (unsymmetrised) d/dlogits weights for the owned rows of `dr`"""
    m = dr.c
    hmax = dr.hmax.long()
    rows = dr.label_off + torch.arange(dr.r)
    cols = torch.arange(m)
    y = ((cols[None, :] >= rows[:, None]) & (cols[None, :] < hmax[rows][:, None])).double()
    bce = torch.clamp(v, min=0) - v * y + torch.log1p(torch.exp(-v.abs()))
    npos = (hmax[rows] - rows).double()
    nneg = m - npos
    per = (bce * y).sum(1) / npos + (bce * (1 - y)).sum(1) / nneg
    w = torch.where(y > 0, 1.0 / npos[:, None], 1.0 / nneg[:, None])
    return per, y, w


def reduce_sums(parts, weights, separate=False, out: Optional[torch.Tensor] = None):
    vals = [float(w) * p.double().sum() for p, w in zip(parts, weights)]
    if separate:
        res = torch.stack(vals).float()
        if out is not None:
            out.copy_(res)
            return out
        return res
    res = torch.stack(vals).sum().float()
    if out is not None:
        out.copy_(res.reshape(out.shape))
        return out
    return res


def clip_backward(dirs: Sequence[Direction], d: int, compute: int, scale, upstream, dscale) -> None:
    CALLS["clip_backward"] += 1
    s = scale.double().item()
    up = upstream.double().item()
    for dr in dirs:
        x, y = dr.x[: dr.r], dr.y[: dr.c]
        if dr.y_t is not None:   # (None: packed for the one-kernel backward, include/mmlearn_hip.h mmk_clip_backward_plan)
            assert torch.equal(dr.y_t[:, : dr.c], y.T), "yT must be the transpose of y"
        else:
            assert ON_CHIP_BACKWARD and dr.mode == 0
        t = x @ y.T
        v = s * t
        if dr.mode == 1:
            _, yy, w = _align_rows(dr, v)
            sig = torch.sigmoid(v)
            a_rc = (sig - yy) * w
            # transposed term a_cr for (c, r): needs the weights of every column's own row
            m = dr.c
            hmax = dr.hmax.long()
            rows = dr.label_off + torch.arange(dr.r)
            cols = torch.arange(m)
            y_cr = ((rows[:, None] >= cols[None, :]) & (rows[:, None] < hmax[cols][None, :])).double()
            npos_c = (hmax[cols] - cols).double()
            w_cr = torch.where(y_cr > 0, 1.0 / npos_c[None, :], 1.0 / (m - npos_c)[None, :])
            g = a_rc + (sig - y_cr) * w_cr
            if dscale is not None:
                dscale += float(up * dr.ds_kappa * (a_rc * t).sum())
            dx = (up * dr.kappa * s) * (g @ y)[:, :d]
            rws = torch.arange(dr.r)
            if dr.dx_accumulate:
                dr.dx.index_add_(0, rws, dx.to(dr.dx.dtype))
            else:
                dr.dx[rws] = dx.to(dr.dx.dtype)
            continue
        p_row = torch.exp(v - dr.lse.double()[:, None])
        p_col = torch.exp(v - dr.lse_col.double()[None, :]) if (dr.c_col or dr.s_col) else torch.zeros_like(v)
        delta = torch.zeros_like(v)
        delta[torch.arange(dr.r), dr.label_off + torch.arange(dr.r)] = 1.0
        g = dr.c_row * p_row + dr.c_col * p_col - dr.c_diag * delta
        gs = dr.s_row * p_row + dr.s_col * p_col - dr.s_diag * delta
        if dscale is not None:
            dscale += float(up * dr.ds_kappa * (gs * t).sum())
        dx = (up * dr.kappa * s) * (g @ y)[:, :d]
        rows = dr.dx_rows.long() if dr.dx_rows is not None else torch.arange(dr.r)
        if dr.normalize:
            src = dr.src.double()[rows]
            nrm = src.norm(dim=-1, keepdim=True)
            inv = 1.0 / nrm.clamp_min(1e-12)
            yv = src * inv
            dx = (dx - yv * (yv * dx).sum(-1, keepdim=True)) * inv
        if dr.dx_accumulate:
            dr.dx.index_add_(0, rows, dx.to(dr.dx.dtype))
        else:
            dr.dx[rows] = dx.to(dr.dx.dtype)


def l2norm_fwd(x, twin=False):
    inv = 1.0 / x.norm(dim=-1).clamp_min(1e-12)
    y = x * inv[..., None]
    return (y, inv, y.bfloat16()) if twin else (y, inv)


def l2norm_bwd(x, dy, inv):
    y = x * inv[..., None]
    return (dy - y * (y * dy).sum(-1, keepdim=True)) * inv[..., None]
