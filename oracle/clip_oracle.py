"""ORACLE (test infrastructure, NOT product code): numpy restatement of the
reference's contrastive-loss path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product (``mmlearn_amd``) never does.

Pinned against golden vectors produced by running the reference itself
(``tests/golden/g1_g2_clip.npz``, ``g3_clip_dist.npz``, ``g4_match.npz``; see
``tests/test_oracle_golden.py``).  One function on the path is third-party and
absent from ``/root/reference``: ``torchmetrics.utilities.compute._safe_matmul``
(torchmetrics 1.6.2, ``uv.lock:2979-2980``).  Its published contract
(``x @ y.T``, fp16 inputs up-cast to fp32) is restated in ``_safe_matmul``
below -- parity unpinned at that single function, everything around it is
pinned through the reference's own outputs.

All arithmetic is float64 unless ``dtype`` says otherwise; gradients are the
closed forms (no autograd), so the oracle is an independent derivation:

    S = s * fa @ fb.T,  R = number of matched pairs
    L = w/2 * ( mean_i(lse_j S_ij - S_ii) + mean_j(lse_i S_ij - S_jj) )
    dL/dS = w/(2R) * (softmax_rows(S) + softmax_cols(S) - 2 I)
"""

from __future__ import annotations

import numpy as np


# ----------------------------------------------------------------------------
# mmlearn/datasets/core/example.py:101-166
def find_matching_indices(first_ids: np.ndarray, second_ids: np.ndarray):
    """All (i, j) with first_ids[i] == second_ids[j], in row-major order
    (``torch.where`` order, example.py:160-166); duplicates give several pairs."""
    if not isinstance(first_ids, np.ndarray) or not isinstance(second_ids, np.ndarray):
        raise TypeError("Expected inputs to be arrays")
    for name, x in (("first_example_ids", first_ids), ("second_example_ids", second_ids)):
        if not (x.ndim == 2 and x.shape[1] == 2):
            raise ValueError(f"Expected argument `{name}` to be a tensor of shape (N, 2), but got shape {x.shape}.")
    m = np.all(first_ids[:, None, :] == second_ids[None, :, :], axis=-1)
    ia, ib = np.nonzero(m)
    return ia.astype(np.int64), ib.astype(np.int64)


# torch.nn.functional.normalize(p=2, dim=-1, eps=1e-12), used at
# tasks/contrastive_pretraining.py:428-429 and losses/contrastive.py:89-90
def l2_normalize(x: np.ndarray, eps: float = 1e-12):
    n = np.sqrt((x * x).sum(-1, keepdims=True))
    return x / np.maximum(n, eps)


def l2_normalize_bwd(x: np.ndarray, dy: np.ndarray, eps: float = 1e-12):
    n = np.sqrt((x * x).sum(-1, keepdims=True))
    nc = np.maximum(n, eps)
    y = x / nc
    # for n > eps: dx = (dy - y (y.dy)) / n ; for clamped rows the norm is a constant
    dx = np.where(n > eps, (dy - y * (y * dy).sum(-1, keepdims=True)) / nc, dy / nc)
    return dx


# torchmetrics 1.6.2 utilities/compute.py::_safe_matmul (contract restated)
def _safe_matmul(x: np.ndarray, y: np.ndarray):
    return x @ y.T


def _lse(x: np.ndarray, axis: int):
    m = x.max(axis=axis, keepdims=True)
    return (m + np.log(np.exp(x - m).sum(axis=axis, keepdims=True))).squeeze(axis)


def cross_entropy_rows(logits: np.ndarray, labels: np.ndarray):
    """F.cross_entropy(logits, labels) (mean) and d/dlogits."""
    r = logits.shape[0]
    lse = _lse(logits, 1)
    loss = (lse - logits[np.arange(r), labels]).mean()
    p = np.exp(logits - lse[:, None])
    p[np.arange(r), labels] -= 1.0
    return loss, p / r


# ----------------------------------------------------------------------------
# losses/contrastive.py:344-413 (_compute_modality_alignment_loss)
def alignment_hmax(sizes):
    """Positive set of row r is {r} + (r, hmax[r]): every modality block k contributes the strict upper triangle
    of [o_k, o_k + n_k) with the reference's NON-cumulative offsets o_0 = 0, o_k = n_{k-1} (quirk Q2, :374-386)."""
    m = int(sum(sizes))
    hmax = np.arange(1, m + 1)
    for k, n in enumerate(sizes):
        o = 0 if k == 0 else sizes[k - 1]
        rows = np.arange(o, o + n)
        hmax[rows] = np.maximum(hmax[rows], o + n)
    return hmax


def alignment_loss(feats, scale):
    """feats: list of [n_k, D] arrays in the reference's dict order.  Returns loss, d/dfeats (list), d/dscale."""
    f = np.concatenate(feats, 0)
    sizes = [len(x) for x in feats]
    m = len(f)
    hmax = alignment_hmax(sizes)
    c = np.arange(m)
    target = ((c[None, :] >= c[:, None]) & (c[None, :] < hmax[:, None])).astype(f.dtype)
    t = _safe_matmul(f, f)
    v = scale * t
    bce = np.maximum(v, 0) - v * target + np.log1p(np.exp(-np.abs(v)))
    num_pos = target.sum(1)
    num_neg = m - num_pos
    loss = ((bce * target).sum(1) / num_pos + (bce * (1 - target)).sum(1) / num_neg).mean()
    w = np.where(target > 0, 1.0 / num_pos[:, None], 1.0 / num_neg[:, None]) / m
    dv = (1.0 / (1.0 + np.exp(-v)) - target) * w
    df = scale * ((dv + dv.T) @ f)
    ds = (dv * t).sum()
    out, o = [], 0
    for n in sizes:
        out.append(df[o:o + n])
        o += n
    return float(loss), out, float(ds)


# ----------------------------------------------------------------------------
# losses/contrastive.py:59-160 with world_size == 1
def contrastive_loss(embeddings: dict, example_ids: dict, scale: float, pairs, l2norm: bool = False,
                     dtype=np.float64, modality_alignment: bool = False):
    """Single-process ContrastiveLoss.forward.

    embeddings: {modality_name: [B_m, D]}, example_ids: {modality_name: int64[B_m, 2]},
    pairs: list of ((mod_a, mod_b), weight).
    Returns dict(loss, grads={mod: dL/dembedding}, dscale, has_graph).
    """
    emb = {k: np.asarray(v, dtype=dtype) for k, v in embeddings.items()}
    raw = emb
    if l2norm:
        emb = {k: l2_normalize(v) for k, v in emb.items()}
    grads = {k: np.zeros_like(v) for k, v in emb.items()}
    loss, dscale, n_terms = 0.0, 0.0, 0
    for (ma, mb), w in pairs:
        if ma not in emb or mb not in emb:  # contrastive.py:266-274
            continue
        ia, ib = find_matching_indices(np.asarray(example_ids[ma]), np.asarray(example_ids[mb]))
        if ia.size == 0:  # :283-287
            continue
        fa, fb = emb[ma][ia], emb[mb][ib]
        r = ia.size
        t = _safe_matmul(fa, fb)  # [R, R]
        logits = scale * t
        labels = np.arange(r)
        la, ga = cross_entropy_rows(logits, labels)
        lb, gb = cross_entropy_rows(logits.T, labels)  # :339-340 (b @ a.T == (a @ b.T).T)
        loss += (la + lb) / 2 * w
        g = (ga + gb.T) * (w / 2)  # dL/dlogits
        dscale += (g * t).sum()
        np.add.at(grads[ma], ia, scale * (g @ fb))
        np.add.at(grads[mb], ib, scale * (g.T @ fa))
        n_terms += 1
    if modality_alignment:  # :146-149, over the embeddings dict in insertion order
        order = list(emb)
        la, dfs, ds = alignment_loss([emb[k] for k in order], scale)
        loss += la
        dscale += ds
        for k, g in zip(order, dfs):
            grads[k] += g
        n_terms += 1
    if l2norm:
        grads = {k: l2_normalize_bwd(raw[k], grads[k]) for k in grads}
    return {"loss": float(loss), "grads": grads, "dscale": float(dscale), "has_graph": n_terms > 0}


# ----------------------------------------------------------------------------
# losses/contrastive.py with world_size > 1, restated for all ranks at once
def contrastive_loss_dist(rank_embeddings, rank_ids, scale: float, pairs, local_loss: bool,
                          gather_with_grad: bool, l2norm: bool = False, dtype=np.float64, modality_alignment: bool = False):
    """Per-rank results of ContrastiveLoss.forward under torch.distributed.

    rank_embeddings[r] / rank_ids[r] are rank r's dicts.  Returns a list (one
    entry per rank) of dict(loss, grads, dscale, has_graph), reproducing the
    four (local_loss, gather_with_grad) cells of contrastive.py:92-110,221-342,
    431-499.
    """
    W = len(rank_embeddings)
    emb_raw = [{k: np.asarray(v, dtype=dtype) for k, v in e.items()} for e in rank_embeddings]
    emb = [{k: l2_normalize(v) for k, v in e.items()} for e in emb_raw] if l2norm else emb_raw
    ids = [{k: np.asarray(v) for k, v in e.items()} for e in rank_ids]
    # _gather_dicts: key union (sorted), placeholder shards dropped, rank order (:463-497)
    keys = sorted({k for e in emb for k in e})
    all_emb, all_ids, owner, local_row = {}, {}, {}, {}
    for k in keys:
        parts = [(r, emb[r][k]) for r in range(W) if k in emb[r]]
        all_emb[k] = np.concatenate([p for _, p in parts], 0)
        all_ids[k] = np.concatenate([ids[r][k] for r, _ in parts], 0)
        owner[k] = np.concatenate([np.full(len(p), r) for r, p in parts])
        local_row[k] = np.concatenate([np.arange(len(p)) for _, p in parts])

    out = [{"loss": 0.0, "grads": {k: np.zeros_like(v) for k, v in emb[r].items()}, "dscale": 0.0,
            "has_graph": False} for r in range(W)]

    for (ma, mb), w in pairs:
        if ma not in all_emb or mb not in all_emb:
            continue
        ia, ib = find_matching_indices(all_ids[ma], all_ids[mb])
        if ia.size == 0:
            continue
        fa_g, fb_g = all_emb[ma][ia], all_emb[mb][ib]
        Rg = ia.size
        if not local_loss:
            t = _safe_matmul(fa_g, fb_g)
            logits = scale * t
            la, ga = cross_entropy_rows(logits, np.arange(Rg))
            lb, gb = cross_entropy_rows(logits.T, np.arange(Rg))
            g = (ga + gb.T) * (w / 2)
            dfa, dfb = scale * (g @ fb_g), scale * (g.T @ fa_g)
            mult = float(W) if gather_with_grad else 1.0  # reduce-scatter of W identical copies
            for r in range(W):
                out[r]["loss"] += (la + lb) / 2 * w
                out[r]["dscale"] += (g * t).sum()
                out[r]["has_graph"] = True
                sel = owner[ma][ia] == r
                if ma in emb[r]:
                    np.add.at(out[r]["grads"][ma], local_row[ma][ia[sel]], mult * dfa[sel])
                sel = owner[mb][ib] == r
                if mb in emb[r]:
                    np.add.at(out[r]["grads"][mb], local_row[mb][ib[sel]], mult * dfb[sel])
        else:
            # local rows of every rank (:276-301); skip_flag ranks contribute size 0 (:196-212)
            loc = []
            for r in range(W):
                if ma in emb[r] and mb in emb[r]:
                    la_, lb_ = find_matching_indices(ids[r][ma], ids[r][mb])
                else:
                    la_ = lb_ = np.zeros(0, np.int64)
                loc.append((la_, lb_))
            sizes = np.array([len(x[0]) for x in loc])
            offs = np.concatenate([[0], np.cumsum(sizes)])
            dfa_g_tot = np.zeros_like(fa_g)
            dfb_g_tot = np.zeros_like(fb_g)
            for r in range(W):
                la_, lb_ = loc[r]
                if la_.size == 0:
                    continue  # reference returns a graph-less 0.0 here (Q3)
                fa_l, fb_l = emb[r][ma][la_], emb[r][mb][lb_]
                labels = offs[r] + np.arange(la_.size)
                ta, tb = _safe_matmul(fa_l, fb_g), _safe_matmul(fb_l, fa_g)
                l_a, g_a = cross_entropy_rows(scale * ta, labels)
                l_b, g_b = cross_entropy_rows(scale * tb, labels)
                g_a, g_b = g_a * (w / 2), g_b * (w / 2)
                out[r]["loss"] += (l_a + l_b) / 2 * w
                out[r]["dscale"] += (g_a * ta).sum() + (g_b * tb).sum()
                out[r]["has_graph"] = True
                np.add.at(out[r]["grads"][ma], la_, scale * (g_a @ fb_g))
                np.add.at(out[r]["grads"][mb], lb_, scale * (g_b @ fa_g))
                dfb_g_tot += scale * (g_a.T @ fa_l)
                dfa_g_tot += scale * (g_b.T @ fb_l)
            if gather_with_grad:  # gathered shards carry grad; backward sums over ranks
                for r in range(W):
                    sel = owner[ma][ia] == r
                    if ma in emb[r]:
                        np.add.at(out[r]["grads"][ma], local_row[ma][ia[sel]], dfa_g_tot[sel])
                    sel = owner[mb][ib] == r
                    if mb in emb[r]:
                        np.add.at(out[r]["grads"][mb], local_row[mb][ib[sel]], dfb_g_tot[sel])
    if modality_alignment:
        # every rank evaluates the same term on the gathered dict (sorted keys, :466); gradients reach the local
        # rows only where the gathered shards carry grad: own shard re-inserted (F,F) x1, dist_nn gather (.,T) xW,
        # none in (T,F) (:491-492)
        la, dfs, ds = alignment_loss([all_emb[k] for k in keys], scale)
        mult = float(W) if gather_with_grad else (0.0 if local_loss else 1.0)
        for r in range(W):
            out[r]["loss"] += la
            out[r]["dscale"] += ds
            out[r]["has_graph"] = True
            for k, g in zip(keys, dfs):
                if k in emb[r]:
                    sel = owner[k] == r
                    out[r]["grads"][k][local_row[k][sel]] += mult * g[sel]
    if l2norm:
        for r in range(W):
            out[r]["grads"] = {k: l2_normalize_bwd(emb_raw[r][k], g) for k, g in out[r]["grads"].items()}
    return out
