"""Tensor-level wrappers over the C ABI (``include/mmlearn_hip.h``).

Each function takes/returns torch tensors that live on the MI355X and enqueues HIP
kernels on the current stream.  torch is used for device memory and streams only; no
ATen math runs here.  There is deliberately no CPU path: CPU tensors raise.
"""

from __future__ import annotations

import contextlib
import ctypes as C
import functools
import threading
import os
from dataclasses import dataclass, field
from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import COMPUTE_BF16, COMPUTE_F32, ClipDir, check, dtype_tag, ptr, require_gpu, stream

MAX_DIRS_PER_CALL = 8

# Test / measurement seams (module attributes, not environment variables: nothing outside this process can flip them).
# All three keep results right to rounding.
BOUNDED_SOFTMAX = True    # False: never hand the row norms to the tile kernels -> per-row / per-column maxima everywhere
PAIR_MIRRORS = True       # False: the two directions of a pair on one rank as two tile passes (round-1 behaviour)
TN_MIN_ROWS = 2048        # from this many rows a mirrored pair stores G once; the mirror's dX uses the transposed-read kernel
ONE_KERNEL_PAIRS = True   # backward of a mirrored pair of >= TN_MIN_ROWS rows on one rank: when each direction alone would run as one
                          # kernel that keeps G on chip (csrc/clip_bwd.hip), the pair is NOT tied (no shared tile pass, no G, no transposed
                          # operands).  Measured (tools/probes/pair_backward_ab.py, device us fwd + bwd, untied / tied): 2048 rows 80 / 109,
                          # 3072 129 / 153, 4096 154 / 181, 8192 414 / 447; below 2048 rows the tied form has no third launch and wins
                          # (1280: 74 / 68, 1536: 81 / 78).  A/B switch

_SEAMS = ("BOUNDED_SOFTMAX", "PAIR_MIRRORS", "TN_MIN_ROWS", "FUSED_LOSS")
_seam_lock = threading.RLock()


@contextlib.contextmanager
def seams(**values):
    """``with kernels.seams(FUSED_LOSS=False, TN_MIN_ROWS=256): ...`` -- set measurement seams for the duration of a block and put
    the previous values back, under a process-wide lock (tests and tools flip them in processes that may run other threads'
    losses; a bare ``kernels.FUSED_LOSS = False`` in a long-lived process stays set for everybody, for ever)."""
    unknown = set(values) - set(_SEAMS)
    if unknown:
        raise KeyError(f"unknown seam(s) {sorted(unknown)}; known: {_SEAMS}")
    g = globals()
    with _seam_lock:
        saved = {k: g[k] for k in values}
        g.update(values)
        try:
            yield
        finally:
            g.update(saved)


def round_up(a: int, b: int) -> int:
    return (a + b - 1) // b * b


def compute_torch_dtype(compute: int) -> torch.dtype:
    return torch.bfloat16 if compute == COMPUTE_BF16 else torch.float32


# ------------------------------------------------------------------ matching
@dataclass
class Match:
    """Result of the device matcher (find_matching_indices, example.py:101-166)."""

    n: int                      # number of matched pairs R
    identity: bool              # R == n_a == n_b and pair p is (p, p)
    idx_a: Optional[torch.Tensor]  # int32[R] (None when identity)
    idx_b: Optional[torch.Tensor]
    repeats_a: bool = False     # some row of a is in more than one pair
    repeats_b: bool = False


@dataclass
class PendingMatch:
    """A matcher launch whose 16-byte status has not been read yet (``match_ids_launch`` -> ``match_ids_finish``)."""

    ids_a: torch.Tensor
    ids_b: torch.Tensor
    counts: torch.Tensor
    idx_a: torch.Tensor
    idx_b: torch.Tensor
    status: torch.Tensor                      # device int32[4]: total, identity, repeats_a, repeats_b
    status_host: Optional[torch.Tensor] = None  # pinned copy, valid once ``event`` has completed
    event: Optional[torch.cuda.Event] = None
    side_stream: Optional[torch.cuda.Stream] = None


def match_ids_launch(ids_a: torch.Tensor, ids_b: torch.Tensor, read_back_async: bool = False) -> PendingMatch:
    """Enqueue the matcher on the current stream.  With ``read_back_async`` the status is also copied to pinned host
    memory behind an event, so a later ``match_ids_finish`` costs no device drain if the stream has moved on."""
    require_gpu(ids_a, "example_ids")
    require_gpu(ids_b, "example_ids")
    ids_a = ids_a.contiguous()
    ids_b = ids_b.contiguous()
    n_a, n_b = ids_a.shape[0], ids_b.shape[0]
    dev = ids_a.device
    cap = max(n_a, n_b)
    counts = torch.empty(_lib.lib().mmk_match_workspace_ints(n_a, n_b), dtype=torch.int32, device=dev)
    idx_a = torch.empty(cap, dtype=torch.int32, device=dev)
    idx_b = torch.empty(cap, dtype=torch.int32, device=dev)
    status = torch.empty(4, dtype=torch.int32, device=dev)
    check(_lib.lib().mmk_match_ids(ptr(ids_a), n_a, ptr(ids_b), n_b, ptr(counts), ptr(idx_a), ptr(idx_b), cap, ptr(status), stream()))
    pm = PendingMatch(ids_a, ids_b, counts, idx_a, idx_b, status)
    if read_back_async:
        pm.status_host = torch.empty(4, dtype=torch.int32, pin_memory=True)
        pm.status_host.copy_(status, non_blocking=True)
        pm.event = torch.cuda.Event()
        pm.event.record()
    return pm


def match_ids_finish(pm: PendingMatch) -> Match:
    n_a, n_b = pm.ids_a.shape[0], pm.ids_b.shape[0]
    dev = pm.ids_a.device
    if pm.event is not None:
        pm.event.synchronize()     # waits for the matcher only, not for whatever the compute stream is doing
        total, ident, rep_a, rep_b = pm.status_host.tolist()
    else:
        total, ident, rep_a, rep_b = pm.status.tolist()  # the one host sync of the loss path
    idx_a, idx_b = pm.idx_a, pm.idx_b
    if total > idx_a.numel():  # heavy duplication: re-run the fill pass with the exact capacity
        idx_a = torch.empty(total, dtype=torch.int32, device=dev)
        idx_b = torch.empty(total, dtype=torch.int32, device=dev)
        check(_lib.lib().mmk_match_ids(ptr(pm.ids_a), n_a, ptr(pm.ids_b), n_b, ptr(pm.counts), ptr(idx_a), ptr(idx_b), total,
                                       ptr(pm.status), stream()))
    if ident:
        return Match(total, True, None, None)
    return Match(total, False, idx_a[:total], idx_b[:total], bool(rep_a), bool(rep_b))


def match_ids(ids_a: torch.Tensor, ids_b: torch.Tensor) -> Match:
    """All (i, j) with ids_a[i] == ids_b[j] in row-major order; one 16-byte D2H read of the status."""
    require_gpu(ids_a, "example_ids")
    require_gpu(ids_b, "example_ids")
    if ids_a.shape[0] == 0 or ids_b.shape[0] == 0:
        dev = ids_a.device
        return Match(0, False, torch.empty(0, dtype=torch.int32, device=dev), torch.empty(0, dtype=torch.int32, device=dev))
    return match_ids_finish(match_ids_launch(ids_a, ids_b))


# ------------------------------------------------------------------ packing
def pack_rows(src: torch.Tensor, idx: Optional[torch.Tensor], r: int, normalize: bool, compute: int, want_transpose: bool):
    """Gather rows ``src[idx]`` (identity when idx is None), optionally L2-normalise, cast to the
    compute type and zero-pad to [r_pad, k_pad]; optionally also emit the transpose [k_pad, r_pad]."""
    require_gpu(src, "embedding")
    src = src.contiguous()
    n_src, d = src.shape
    r_pad, k_pad = round_up(max(r, 1), 128), round_up(d, 64)
    cdt = compute_torch_dtype(compute)
    dst = torch.empty((r_pad, k_pad), dtype=cdt, device=src.device)
    dst_t = torch.empty((k_pad, r_pad), dtype=cdt, device=src.device) if want_transpose else None
    nrm = torch.empty(r_pad, dtype=torch.float32, device=src.device)
    if idx is not None:
        assert idx.dtype == torch.int32 and idx.numel() >= r
    check(_lib.lib().mmk_pack_rows(ptr(src), dtype_tag(src.dtype), n_src, d, ptr(idx), r, int(normalize), ptr(dst), ptr(dst_t),
                                   r_pad, k_pad, r_pad, compute, ptr(nrm), stream()))
    dst._mmk_norm = nrm   # L2 norms of the packed rows: lets the forward bound the logits of a tile (Direction.x / .y)
    return dst, dst_t


def row_norms(t: torch.Tensor) -> Optional[torch.Tensor]:
    """The norms ``pack_rows`` attached to a packed operand (None for tensors that did not come from it)."""
    return getattr(t, "_mmk_norm", None)


def slice_packed(t: torch.Tensor, p0: int) -> torch.Tensor:
    """Rows ``p0:`` of a packed operand, norms included."""
    out = t[p0:]
    n = row_norms(t)
    if n is not None:
        out._mmk_norm = n[p0:]
    return out


def pack_rows_many(reqs: Sequence[tuple], compute: int) -> list:
    """``[(src, idx, r, normalize, want_transpose), ...]`` -> ``[(dst, dst_t), ...]`` with ONE launch when the operands share
    dtype and width (the two sides of a pair do); otherwise falls back to one ``pack_rows`` each."""
    srcs = [q[0] for q in reqs]
    if len(reqs) < 2 or len(reqs) > MAX_DIRS_PER_CALL or len({(t.dtype, t.shape[1], t.device) for t in srcs}) != 1:
        return [pack_rows(*q[:4], compute, q[4]) for q in reqs]
    cdt = compute_torch_dtype(compute)
    d = srcs[0].shape[1]
    k_pad = round_up(d, 64)
    arr = (_lib.PackReq * len(reqs))()
    out, keep = [], []
    for k, (src, idx, r, normalize, want_t) in enumerate(reqs):
        require_gpu(src, "embedding")
        src = src.contiguous()
        keep.append(src)
        r_pad = round_up(max(r, 1), 128)
        dst = torch.empty((r_pad, k_pad), dtype=cdt, device=src.device)
        dst_t = torch.empty((k_pad, r_pad), dtype=cdt, device=src.device) if want_t else None
        nrm = torch.empty(r_pad, dtype=torch.float32, device=src.device)
        if idx is not None:
            assert idx.dtype == torch.int32 and idx.numel() >= r
        e = arr[k]
        e.src, e.idx, e.dst, e.dstT, e.r, e.r_pad, e.normalize, e.ldt = ptr(src), ptr(idx), ptr(dst), ptr(dst_t), r, r_pad, int(normalize), r_pad
        e.norm = ptr(nrm)
        dst._mmk_norm = nrm
        out.append((dst, dst_t))
    check(_lib.lib().mmk_pack_rows_many(C.cast(arr, C.c_void_p), len(reqs), dtype_tag(srcs[0].dtype), d, k_pad, compute, stream()))
    return out


# ------------------------------------------------------------------ CLIP loss
@dataclass
class Direction:
    """One CE direction: rows of ``s * X @ Y^T`` with label(i) = label_off + i (mmk_clip_dir)."""

    x: torch.Tensor                 # packed owned rows  [>= r, k_pad]
    y: torch.Tensor                 # packed columns     [>= c, k_pad]
    y_t: Optional[torch.Tensor]     # [k_pad, ldt]
    r: int
    c: int
    label_off: int = 0
    kappa: float = 1.0              # w / (2 * rows in the mean) [x W for gather_with_grad]
    ds_kappa: float = 1.0           # factor of the d/dscale reduction
    # gradient recipe: G = c_row*P_row + c_col*P_col - c_diag*delta ; same triple for dscale
    c_row: float = 1.0
    c_col: float = 1.0
    c_diag: float = 2.0
    s_row: float = 1.0
    s_col: float = 1.0
    s_diag: float = 2.0
    # forward outputs
    lse: Optional[torch.Tensor] = None
    diag: Optional[torch.Tensor] = None
    loss_part: Optional[torch.Tensor] = None   # block partial sums of (lse_i - diag_i)
    # backward inputs
    lse_col: Optional[torch.Tensor] = None
    dx: Optional[torch.Tensor] = None        # [n_src, d] gradient buffer
    dx_rows: Optional[torch.Tensor] = None   # int32[r]
    dx_accumulate: bool = False
    src: Optional[torch.Tensor] = None       # original rows when normalize=True
    normalize: bool = False
    mode: int = 0                            # 0: cross-entropy direction, 1: modality-alignment BCE rows
    hmax: Optional[torch.Tensor] = None      # int32[c] (mode 1)
    _keep: list = field(default_factory=list)


@functools.lru_cache(maxsize=256)
def _plan(r: int, c: int, k_pad: int, compute: int):
    """(column tiles, gradient blocks, K splits) of one direction: a pure function of the shape, asked once per shape."""
    a, b, s = C.c_int32(), C.c_int32(), C.c_int32()
    check(_lib.lib().mmk_clip_plan(r, c, k_pad, compute, C.addressof(a), C.addressof(b), C.addressof(s)))
    return a.value, b.value, s.value


@functools.lru_cache(maxsize=256)
def _wgrad_ws_floats(m: int, n: int, k: int) -> int:
    splits, wsf = C.c_int(0), C.c_int64(0)
    check(_lib.lib().mmk_wgrad_plan(m, n, k, C.cast(C.pointer(splits), C.c_void_p), C.cast(C.pointer(wsf), C.c_void_p)))
    return int(wsf.value)


def _mirror_of(a: Direction, b: Direction, backward: bool = False) -> bool:
    """``b`` is ``a`` with the operands swapped on one rank: its logits are the transposed logits of ``a``
    (``logits_per_feature_b = logits_per_feature_a.T``, contrastive.py:327-340), so one pass over the similarity tiles
    serves both (row statistics for ``a``, column statistics for ``b``; G for ``a``, G^T for ``b``)."""
    ok = (a.mode == 0 and b.mode == 0 and a.r == a.c == b.r == b.c and a.label_off == 0 and b.label_off == 0
          and a.x.data_ptr() == b.y.data_ptr() and a.y.data_ptr() == b.x.data_ptr() and a.x.shape == b.y.shape and a.y.shape == b.x.shape)
    if ok and backward:   # G_b = G_a^T needs the same row / column weights in both directions
        ok = (a.c_row == a.c_col == b.c_row == b.c_col and a.c_diag == b.c_diag and b.s_row == b.s_col == b.s_diag == 0.0
              and a.lse_col is not None and b.lse is not None and a.lse_col.data_ptr() == b.lse.data_ptr()
              and b.lse_col is not None and b.lse_col.data_ptr() == a.lse.data_ptr())
    return ok


def _pair_up(dirs: Sequence[Direction], backward: bool = False) -> list:
    """[(direction, mirror or None)]: consecutive mirrored directions share one tile pass (``PAIR_MIRRORS = False``: A/B switch)."""
    out, k = [], 0
    no = not PAIR_MIRRORS
    while k < len(dirs):
        if not no and k + 1 < len(dirs) and _mirror_of(dirs[k], dirs[k + 1], backward):
            out.append((dirs[k], dirs[k + 1]))
            k += 2
        else:
            out.append((dirs[k], None))
            k += 1
    return out


_TICKETS: dict = {}


def _tn_min_rows() -> int:
    """Mirrored pairs with at least this many rows store G once and form the second direction's gradient with the
    transposed-read kernel; smaller ones store G and G^T from the tile pass (one launch fewer)."""
    return int(TN_MIN_ROWS)


def _tickets(dev: torch.device, n: int) -> torch.Tensor:
    """The workspace of the in-launch loss combine: slot 0 is a ticket counter that the merge launch expects at zero and
    leaves at zero, so ONE buffer per (device, stream) is reused call after call (launches of a stream run one after the
    other); it is dropped whenever a call fails, because an aborted launch may leave a count behind."""
    key = (dev.index, stream())
    t = _TICKETS.get(key)
    if t is None or t.numel() < n:
        t = torch.zeros(max(n, 1024), dtype=torch.int32, device=dev)
        _TICKETS[key] = t
    return t


def clip_forward(dirs: Sequence[Direction], d: int, compute: int, scale: torch.Tensor,
                 loss_weights: Optional[Sequence[float]] = None) -> Optional[torch.Tensor]:
    """Fills dir.lse / dir.diag / dir.loss_part for every direction (two launches for all of them: similarity tiles with
    their statistics, merge of the tile partials).  The two directions of a pair on one rank are computed from ONE pass over
    the similarity tiles (see ``_mirror_of``).  With ``loss_weights`` (one per direction) the merge launch also forms the
    loss value ``sum_k w_k * sum_i (lse_i - diag_i)``: the 0-dim tensor is returned; ``None`` means the
    caller combines ``loss_part`` with ``reduce_sums`` (more directions than one call takes, or alignment rows)."""
    assert scale.dtype == torch.float32 and scale.is_cuda
    dev = scale.device
    pairs = _pair_up(dirs)
    fuse_loss = loss_weights is not None and len(pairs) <= MAX_DIRS_PER_CALL and all(x.mode == 0 for x in dirs)
    w_of = {id(x): float(w) for x, w in zip(dirs, loss_weights)} if fuse_loss else {}
    loss_out = None
    for i0 in range(0, len(pairs), MAX_DIRS_PER_CALL):
        chunk = pairs[i0:i0 + MAX_DIRS_PER_CALL]
        arr = (ClipDir * len(chunk))()
        k_pad = chunk[0][0].x.shape[1]
        for k, (dr, mir) in enumerate(chunk):
            assert dr.x.shape[1] == k_pad and dr.y.shape[1] == k_pad
            n_col_tiles, _, _ = _plan(dr.r, dr.c, k_pad, compute)
            part = torch.empty((n_col_tiles, dr.r, 2), dtype=torch.float32, device=dev)
            if dr.mode == 0:
                dr.lse = torch.empty(dr.r, dtype=torch.float32, device=dev)
                dr.diag = torch.empty(dr.r, dtype=torch.float32, device=dev)
            # mode 0: one entry per block of 64 rows; mode 1: per 256 rows
            dr.loss_part = torch.empty((dr.r + 63) // 64 if dr.mode == 0 else (dr.r + 255) // 256, dtype=torch.float32, device=dev)
            dr._keep.append(part)
            e = arr[k]
            e.x, e.y, e.r, e.c, e.label_off = ptr(dr.x), ptr(dr.y), dr.r, dr.c, dr.label_off
            e.part, e.diag, e.lse, e.loss_part = ptr(part), ptr(dr.diag), ptr(dr.lse), ptr(dr.loss_part)
            e.mode, e.hmax = dr.mode, ptr(dr.hmax)
            xn, yn = (row_norms(dr.x), row_norms(dr.y)) if BOUNDED_SOFTMAX else (None, None)
            if xn is not None and yn is not None and dr.mode == 0:
                assert xn.numel() >= dr.r and yn.numel() >= dr.c
                e.x_norm, e.y_norm = ptr(xn), ptr(yn)
                dr._keep += [xn, yn]
            if mir is not None:
                mpart = torch.empty((_lib.lib().mmk_clip_mirror_tiles(dr.r), dr.c, 2), dtype=torch.float32, device=dev)
                mir.lse = torch.empty(mir.r, dtype=torch.float32, device=dev)
                mir.diag = dr.diag          # label_off = 0 on both sides: the same diagonal
                mir.loss_part = torch.empty((mir.r + 63) // 64, dtype=torch.float32, device=dev)
                mir._keep.append(mpart)
                e.mirror_part, e.mirror_lse, e.mirror_loss_part = ptr(mpart), ptr(mir.lse), ptr(mir.loss_part)
        n_tick = _lib.lib().mmk_clip_tickets(C.cast(arr, C.c_void_p), len(chunk))
        tickets = _tickets(dev, n_tick)
        try:
            if fuse_loss:
                ws = [w_of[id(x)] for pair in chunk for x in pair if x is not None]
                w_arr = (C.c_float * len(ws))(*ws)
                loss_out = torch.empty((), dtype=torch.float32, device=dev)
                check(_lib.lib().mmk_clip_forward_loss(C.cast(arr, C.c_void_p), len(chunk), k_pad, d, compute, ptr(scale), C.cast(w_arr, C.c_void_p),
                                                       ptr(loss_out), ptr(tickets), tickets.numel(), stream()))
            else:
                check(_lib.lib().mmk_clip_forward(C.cast(arr, C.c_void_p), len(chunk), k_pad, d, compute, ptr(scale), ptr(tickets), tickets.numel(),
                                                  stream()))
        except Exception:
            _TICKETS.pop((dev.index, stream()), None)
            raise
    return loss_out


def reduce_sums(parts: Sequence[torch.Tensor], weights: Sequence[float], separate: bool = False,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """separate=False: 0-dim ``sum_k weights[k] * parts[k].sum()`` (the loss value, contrastive.py:134-144,160);
    separate=True: ``out[k] = weights[k] * parts[k].sum()``.  One tiny launch."""
    n = len(parts)
    if n > 2 * MAX_DIRS_PER_CALL:
        raise ValueError("too many terms for one reduce_sums call")
    dev = parts[0].device
    if out is None:
        out = torch.empty(n if separate else (), dtype=torch.float32, device=dev)
    ptrs = (C.c_void_p * n)(*[ptr(t) for t in parts])
    cnts = (C.c_int32 * n)(*[t.numel() for t in parts])
    ws = (C.c_float * n)(*[float(w) for w in weights])
    check(_lib.lib().mmk_reduce_sums(C.cast(ptrs, C.c_void_p), C.cast(cnts, C.c_void_p), C.cast(ws, C.c_void_p), n, int(separate),
                                     ptr(out), stream()))
    return out


def _backward_plan(shapes: Sequence[tuple], k_pad: int, compute: int) -> list:
    return list(_backward_plan_cached(tuple(shapes), k_pad, compute))


@functools.lru_cache(maxsize=256)
def _backward_plan_cached(shapes: tuple, k_pad: int, compute: int) -> tuple:
    """``[(r, c, mode, mirror_of), ...]`` (``mirror_of``: index of the direction whose gradient tiles this one reuses, or None) ->
    per direction, whether ``mmk_clip_backward`` runs it as ONE kernel that recomputes its tiles on chip (csrc/clip_bwd.hip): such a
    direction needs neither the transposed operand nor the G workspace.  A host-side query, no buffers involved."""
    arr = (ClipDir * len(shapes))()
    for k, (r, c, mode, mirror_of) in enumerate(shapes):
        e = arr[k]
        e.r, e.c, e.mode = r, c, mode
        if mirror_of is not None:   # described as in the real call: same g, g_ready on the second; only compared, never read
            e.g_ready = 1
            e.g = arr[mirror_of].g = C.c_void_p(mirror_of + 1)
    flags = (C.c_int32 * len(shapes))()
    check(_lib.lib().mmk_clip_backward_plan(C.cast(arr, C.c_void_p), len(shapes), k_pad, compute, C.cast(flags, C.c_void_p)))
    return tuple(bool(f) for f in flags)


def backward_recomputes_on_chip(r: int, c: int, d: int, compute: int, n_dirs: int = 2) -> bool:
    """Would ``n_dirs`` unpaired cross-entropy directions of r owned rows x c columns each take the one-kernel backward?  What
    ``losses`` asks before it packs the gathered operands: such directions need no transposed copy (``clip_backward`` makes one
    on demand if the answer turns out different for the directions it is finally given)."""
    return all(_backward_plan([(r, c, 0, None)] * n_dirs, round_up(d, 64), compute))


def pair_runs_untied(n: int, d: int, compute: int, n_dirs: int = 2) -> bool:
    """A mirrored pair of n x n directions on one rank: will ``clip_backward`` run its halves as two one-kernel directions?"""
    return bool(ONE_KERNEL_PAIRS) and n >= _tn_min_rows() and backward_recomputes_on_chip(n, n, d, compute, n_dirs)


def transposed_operand(y: torch.Tensor, c: int, compute: int) -> torch.Tensor:
    """[k_pad, c_pad] transpose of a packed operand that was packed without one."""
    return pack_rows(y, None, c, False, compute, True)[1]


def clip_backward(dirs: Sequence[Direction], d: int, compute: int, scale: torch.Tensor, upstream: torch.Tensor,
                  dscale: Optional[torch.Tensor]) -> None:
    """dX for every direction (scattered into dir.dx) and dscale += d loss / d scale.  Mirrored directions share one
    gradient-tile pass: it stores G for the first and G^T, which is the second one's G.  Row-sharded directions run as one kernel
    each that keeps G on chip (``_backward_plan``): no G workspace, ``y_t`` may be None."""
    dev = scale.device
    assert upstream.dtype == torch.float32 and upstream.is_cuda
    cdt = compute_torch_dtype(compute)
    mirror_src = {}
    for a, b in _pair_up(dirs, backward=True):
        if b is not None:
            mirror_src[id(b)] = a
    if mirror_src and ONE_KERNEL_PAIRS:
        for i0 in range(0, len(dirs), MAX_DIRS_PER_CALL):
            chunk = dirs[i0:i0 + MAX_DIRS_PER_CALL]
            alone = _backward_plan([(x.r, x.c, x.mode, None) for x in chunk], chunk[0].x.shape[1], compute)
            for k, x in enumerate(chunk):   # (the two halves of a pair have the same shape: flagged together or not at all)
                if alone[k] and x.r >= _tn_min_rows() and id(x) in mirror_src:
                    del mirror_src[id(x)]
    for i0 in range(0, len(dirs), MAX_DIRS_PER_CALL):
        chunk = dirs[i0:i0 + MAX_DIRS_PER_CALL]
        if any(id(x) in mirror_src for x in chunk[:1]):   # a pair must not straddle two calls
            mirror_src.pop(id(chunk[0]), None)
        arr = (ClipDir * len(chunk))()
        k_pad = chunk[0].x.shape[1]
        r_max, c_max = max(x.r for x in chunk), max(x.c for x in chunk)
        _, _, n_split = _plan(r_max, c_max, k_pad, compute)
        keep = []
        gbuf = {}
        in_chunk = {id(x) for x in chunk}
        pos = {id(x): k for k, x in enumerate(chunk)}
        on_chip = _backward_plan([(x.r, x.c, x.mode, pos.get(id(mirror_src.get(id(x))))) for x in chunk], k_pad, compute)
        for k, dr in enumerate(chunk):
            r_pad, c_pad = round_up(dr.r, 128), round_up(dr.c, 128)
            _, n_grad_blocks, _ = _plan(dr.r, dr.c, k_pad, compute)
            src = mirror_src.get(id(dr))
            ready = src is not None and id(src) in gbuf
            tn = ready and gbuf[id(src)][1] is None     # the mirror source kept G only: this direction reads it transposed
            if tn:
                g = gbuf[id(src)][0]
            elif on_chip[k]:
                g = None
            else:
                g = gbuf[id(src)][1] if ready else torch.empty((r_pad, c_pad), dtype=cdt, device=dev)
            if dr.y_t is None and not on_chip[k]:   # packed on the expectation of the one-kernel backward, which this call does not take
                dr.y_t = transposed_operand(dr.y, dr.c, compute)
            slab = torch.empty((1 if tn else n_split, 1 if tn else r_pad, k_pad), dtype=torch.float32, device=dev)
            ds_part = torch.empty(n_grad_blocks, dtype=torch.float32, device=dev)
            keep += [g, slab, ds_part]
            e = arr[k]
            e.x, e.y, e.yT, e.r, e.c, e.label_off = ptr(dr.x), ptr(dr.y), ptr(dr.y_t), dr.r, dr.c, dr.label_off
            e.ldt = dr.y_t.shape[1] if dr.y_t is not None else c_pad
            e.lse, e.lse_col, e.g, e.ldg = ptr(dr.lse), ptr(dr.lse_col), ptr(g), c_pad
            e.c_row, e.c_col, e.c_diag = dr.c_row, dr.c_col, dr.c_diag
            e.s_row, e.s_col, e.s_diag = dr.s_row, dr.s_col, dr.s_diag
            e.kappa, e.ds_kappa, e.slab, e.ds_part = dr.kappa, dr.ds_kappa, ptr(slab), ptr(ds_part)
            e.dx, e.dx_rows, e.dx_dtype = ptr(dr.dx), ptr(dr.dx_rows), dtype_tag(dr.dx.dtype)
            e.dx_accumulate = int(dr.dx_accumulate)
            e.src, e.normalize = ptr(dr.src), int(dr.normalize)
            e.mode, e.hmax = dr.mode, ptr(dr.hmax)
            xn, yn = (row_norms(dr.x), row_norms(dr.y)) if BOUNDED_SOFTMAX else (None, None)
            if xn is not None and yn is not None and dr.mode == 0:
                e.x_norm, e.y_norm = ptr(xn), ptr(yn)
                keep += [xn, yn]
            e.src_dtype = dtype_tag(dr.src.dtype) if dr.src is not None else 0
            e.g_ready = int(ready)
            if tn:
                wsf = _wgrad_ws_floats(dr.c, r_pad, k_pad)
                tn_ws = torch.empty(wsf, dtype=torch.float32, device=dev)
                keep.append(tn_ws)
                e.g_transposed, e.tn_ws, e.tn_ws_floats = 1, ptr(tn_ws), wsf
                e.ldg = g.shape[1]
            mir = next((b for b in chunk[k + 1:k + 2] if mirror_src.get(id(b)) is dr), None)
            if mir is not None and id(mir) in in_chunk:
                if compute == COMPUTE_BF16 and dr.r >= _tn_min_rows():
                    # large mirrored pair: G is stored once; the mirror's dX = G^T Y comes from the transposed-read kernel
                    gbuf[id(dr)] = (g, None)
                else:
                    gt = torch.empty((c_pad, r_pad), dtype=cdt, device=dev)     # = the mirror's [r_pad', c_pad'] G
                    gbuf[id(dr)] = (g, gt)
                    keep.append(gt)
                    e.gT, e.ldgt = ptr(gt), r_pad
        check(_lib.lib().mmk_clip_backward(C.cast(arr, C.c_void_p), len(chunk), k_pad, d, compute, ptr(scale), ptr(upstream),
                                           ptr(dscale), stream()))
        del keep  # the caching allocator keeps the blocks alive until the stream has consumed them


# ------------------------------------------------------------------ CLIP loss: one resident-grid launch (small batches)
FUSED_MAX_ROWS = 1024     # matched rows per pair
FUSED_MAX_PAIRS = 4
FUSED_LOSS = True         # measurement seam: False sends every call down the tiled multi-launch path


class _FusedPlan:
    """Per (device, rows per pair, d, dtype): workspace size, grid and the co-residency verdict of ``mmk_clip_fused_plan`` (one
    ctypes call, once), plus a pool of zero-initialised workspaces.  A workspace is checked out by a forward call and returns
    to the pool when its backward has run (or the forward's handle is dropped), so two losses alive at once never share
    raw-gradient storage; the counters inside are zero again whenever a launch has completed, so reuse needs no memset."""

    __slots__ = ("ws_bytes", "grid", "capacity", "pool", "device", "allocs")

    def __init__(self, device: torch.device, ns: tuple, d: int, dtype: torch.dtype):
        n_arr = (C.c_int32 * len(ns))(*ns)
        wsb, grid, cap = C.c_int64(0), C.c_int32(0), C.c_int32(0)
        check(_lib.lib().mmk_clip_fused_plan(C.cast(n_arr, C.c_void_p), len(ns), d, dtype_tag(dtype), C.addressof(wsb), C.addressof(grid),
                                             C.addressof(cap)))
        self.ws_bytes, self.grid, self.capacity = wsb.value, grid.value, cap.value
        self.pool: list = []
        self.device = device
        self.allocs = 0   # workspaces ever allocated (a steady-state loop needs one or two)

    def take(self, stream_handle: int) -> torch.Tensor:
        """A workspace last used on THIS stream (its previous launches are ordered before ours), else a fresh zeroed one.  Under
        graph capture always a fresh one, which the run then never returns: the graph bakes the pointer in, so the buffer must be
        owned by the graph (its private pool) and by nobody else -- not by later eager calls on the same stream."""
        if torch.cuda.is_current_stream_capturing():
            return torch.zeros(self.ws_bytes, dtype=torch.uint8, device=self.device)
        for k in range(len(self.pool) - 1, -1, -1):
            if self.pool[k][0] == stream_handle:
                return self.pool.pop(k)[1]
        self.allocs += 1
        return torch.zeros(self.ws_bytes, dtype=torch.uint8, device=self.device)

    def give(self, stream_handle: int, ws: torch.Tensor) -> None:
        if len(self.pool) < 4:   # more than a few idle workspaces per shape are not worth their memory
            self.pool.append((stream_handle, ws))


_FUSED_PLANS: dict = {}
_FUSED_EVENTS: dict = {}   # (device index, stream handle) -> event recorded behind that stream's latest fused launch
_FUSED_LAST: dict = {}     # device index -> (event, stream handle) of the device's latest fused launch


def clip_fused_plan(device: torch.device, ns: Sequence[int], d: int, dtype: torch.dtype) -> Optional[_FusedPlan]:
    """The plan if the one-launch path serves these shapes (and its grid is co-resident on this device), else None."""
    if not FUSED_LOSS or not (0 < len(ns) <= FUSED_MAX_PAIRS) or dtype not in (torch.float32, torch.bfloat16):
        return None
    if d % (4 if dtype == torch.float32 else 8) or any(not (0 < n <= FUSED_MAX_ROWS) for n in ns):
        return None
    # workspace layout, grid and capacity depend on the rows of a pair only through its tile count: one plan (and one workspace
    # pool) per tile-count tuple, so a pairing whose matched-row count changes from batch to batch does not grow the cache
    nts = tuple((int(n) + 63) // 64 for n in ns)
    key = (device.index, nts, d, dtype)
    plan = _FUSED_PLANS.get(key)
    if plan is None:
        plan = _FUSED_PLANS[key] = _FusedPlan(device, tuple(64 * t for t in nts), d, dtype)
    return plan if plan.grid <= plan.capacity else None


class FusedRun:
    """Forward state of one fused call: keeps the workspace (raw gradient sums) until ``backward`` or deletion."""

    __slots__ = ("plan", "ws", "arr", "n_pairs", "d", "dtype", "ds_raw", "ds_acc", "keep", "stream", "n_bwd", "captured")

    def release(self, last_stream: Optional[int] = None) -> None:
        """Hand the workspace back, tagged with the stream its last launch went to (the pool only reuses it on that stream).  A
        workspace taken under graph capture belongs to that graph and is simply dropped here."""
        ws, self.ws = self.ws, None
        if ws is not None and self.plan is not None and not getattr(self, "captured", False):
            self.plan.give(self.stream if last_stream is None else last_stream, ws)

    def __del__(self):
        # forward without backward (evaluation, a dropped graph): the launch that used the workspace is queued on self.stream
        try:
            self.release()
        except Exception:
            pass


def clip_fused_forward(plan: _FusedPlan, pairs: Sequence[tuple], d: int, scale: torch.Tensor, want_grad: bool):
    """``pairs``: [(a, b, idx_a, idx_b, n, weight)] with a / b the contiguous embedding matrices and idx int32 match lists or
    None.  ONE launch: returns (0-dim f32 loss, FusedRun)."""
    dev = scale.device
    arr = (_lib.FusedPair * len(pairs))()
    keep = []
    for k, (a, b, ia, ib, n, w) in enumerate(pairs):
        e = arr[k]
        e.a, e.b, e.idx_a, e.idx_b, e.n, e.weight = ptr(a), ptr(b), ptr(ia), ptr(ib), n, float(w)
        keep += [a, b, ia, ib]
    run = FusedRun()
    run.plan, run.arr, run.n_pairs, run.d, run.dtype, run.keep = plan, arr, len(pairs), d, pairs[0][0].dtype, keep
    run.stream, run.n_bwd = stream(), 0
    run.captured = torch.cuda.is_current_stream_capturing()
    run.ws = plan.take(run.stream)
    out = torch.empty(3, dtype=torch.float32, device=dev)   # [loss, raw d loss / d scale, 0 = accumulator for backward's dscale]
    run.ds_raw, run.ds_acc = out[1:], out[2:]
    # The kernel is a resident grid with in-launch hand-offs: its workgroups must all be on the chip together.  Two such grids
    # launched from different streams can each hold part of the slots and wait for the rest until the bounded spins give up (NaN
    # loss), so fused launches of a device are chained: a launch waits for the previous one if that went to another stream.
    # (Not under capture: a captured launch replays inside its graph, ordered by the graph.)
    if not run.captured:
        cur = torch.cuda.current_stream(dev)
        last = _FUSED_LAST.get(dev.index)
        if last is not None and last[1] != run.stream:
            cur.wait_event(last[0])
    check(_lib.lib().mmk_clip_fused_forward(C.cast(arr, C.c_void_p), len(pairs), d, dtype_tag(run.dtype), ptr(scale), ptr(run.ws), plan.ws_bytes,
                                            int(want_grad), ptr(out), ptr(run.ds_raw) if want_grad else None, run.stream))
    if not run.captured:
        ev = _FUSED_EVENTS.get((dev.index, run.stream))
        if ev is None:
            ev = _FUSED_EVENTS[(dev.index, run.stream)] = torch.cuda.Event()
        ev.record(cur)
        _FUSED_LAST[dev.index] = (ev, run.stream)
    return out[0], run


def clip_fused_backward(run: FusedRun, grads: Sequence[tuple], scale: torch.Tensor, upstream: torch.Tensor, dscale: Optional[torch.Tensor]) -> None:
    """``grads``: [(da, db, accumulate_a, accumulate_b)] per pair (user-dtype buffers, or zeroed f32 ones when accumulating).
    ONE launch.  The raw sums are only read, so a second backward (``retain_graph=True``) repeats it; the workspace returns to the
    pool when the run is dropped."""
    if run.ws is None:
        raise RuntimeError("mmlearn_amd: backward through a one-launch loss whose workspace was already released")
    for k, (da, db, acc_a, acc_b) in enumerate(grads):
        e = run.arr[k]
        e.da, e.db, e.da_accumulate, e.db_accumulate = ptr(da), ptr(db), int(acc_a), int(acc_b)
    dt = grads[0][0].dtype
    assert all(g[0].dtype == dt and g[1].dtype == dt for g in grads)
    check(_lib.lib().mmk_clip_fused_backward(C.cast(run.arr, C.c_void_p), run.n_pairs, run.d, dtype_tag(dt), ptr(scale), ptr(upstream), ptr(run.ws),
                                             run.plan.ws_bytes, ptr(run.ds_raw), ptr(dscale), stream()))
    run.stream = stream()   # the workspace's last launch; it goes back to the pool when the autograd graph lets go of the run


# ------------------------------------------------------------------ row ops
def l2norm_fwd(x: torch.Tensor, twin: bool = False):
    """-> (y, 1 / norm per row[, y rounded to bf16 when ``twin``])"""
    require_gpu(x)
    x2 = x.contiguous().view(-1, x.shape[-1])
    y = torch.empty_like(x2)
    inv = torch.empty(x2.shape[0], dtype=torch.float32, device=x.device)
    if not twin:
        check(_lib.lib().mmk_l2norm_fwd(ptr(x2), ptr(y), ptr(inv), x2.shape[0], x2.shape[1], dtype_tag(x.dtype), stream()))
        return y.view(x.shape), inv
    y16 = torch.empty(x2.shape, dtype=torch.bfloat16, device=x.device)
    check(_lib.lib().mmk_l2norm_fwd_twin(ptr(x2), ptr(y), ptr(y16), ptr(inv), x2.shape[0], x2.shape[1], dtype_tag(x.dtype), stream()))
    return y.view(x.shape), inv, y16.view(x.shape)


def l2norm_bwd(x: torch.Tensor, dy: torch.Tensor, inv: torch.Tensor) -> torch.Tensor:
    x2 = x.contiguous().view(-1, x.shape[-1])
    dy2 = dy.contiguous().view(-1, x.shape[-1])
    dx = torch.empty_like(x2)
    check(_lib.lib().mmk_l2norm_bwd(ptr(x2), ptr(dy2), ptr(inv), ptr(dx), x2.shape[0], x2.shape[1], dtype_tag(x.dtype), stream()))
    return dx.view(x.shape)


# ------------------------------------------------------------------ I-JEPA ops
def mask_to_index(mask: torch.Tensor, keep: int) -> tuple[torch.Tensor, torch.Tensor]:
    """mask int32[b, n] of 0/1 -> sorted keep indices int32[b, keep]; second value is a device flag
    (int32[1]) that is 1 when some row's popcount != keep."""
    require_gpu(mask, "mask")
    mask = mask.to(torch.int32).contiguous()
    b, n = mask.shape
    idx = torch.zeros((b, keep), dtype=torch.int32, device=mask.device)
    bad = torch.zeros(1, dtype=torch.int32, device=mask.device)
    check(_lib.lib().mmk_mask_to_index(ptr(mask), b, n, keep, ptr(idx), ptr(bad), stream()))
    return idx, bad


def gather_rows(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """x [b, n, d], idx int32 [n_masks, b or 1, keep] -> [n_masks*b, keep, d]."""
    require_gpu(x)
    x = x.contiguous()
    b, n, d = x.shape
    n_masks, idx_b, keep = idx.shape
    out = torch.empty((n_masks * b, keep, d), dtype=x.dtype, device=x.device)
    check(_lib.lib().mmk_gather_rows(ptr(x), ptr(out), ptr(idx), b, n, d, n_masks, idx_b, keep, dtype_tag(x.dtype), stream()))
    return out


def scatter_rows(dout: torch.Tensor, idx: torch.Tensor, b: int, n: int) -> torch.Tensor:
    dout = dout.contiguous()
    d = dout.shape[-1]
    n_masks, idx_b, keep = idx.shape
    dx = torch.empty((b, n, d), dtype=dout.dtype, device=dout.device)
    check(_lib.lib().mmk_scatter_rows(ptr(dout), ptr(dx), ptr(idx), b, n, d, n_masks, idx_b, keep, dtype_tag(dout.dtype), stream()))
    return dx


def ijepa_loss_fwd(z: torch.Tensor, h: torch.Tensor, idx: torch.Tensor, kind: int, eps: float, want_target: bool):
    require_gpu(z)
    z, h = z.contiguous(), h.contiguous()
    b, n, d = h.shape
    n_masks, idx_b, keep = idx.shape
    rows = n_masks * b * keep
    assert z.shape == (n_masks * b, keep, d), (z.shape, (n_masks * b, keep, d))
    n_blocks = _lib.lib().mmk_ijepa_loss_blocks(rows)
    part = torch.empty(n_blocks, dtype=torch.float32, device=z.device)
    loss = torch.empty((), dtype=torch.float32, device=z.device)
    target = torch.empty((n_masks * b, keep, d), dtype=h.dtype, device=z.device) if want_target else None
    dt = dtype_tag(z.dtype) | (dtype_tag(h.dtype) << 4)
    check(_lib.lib().mmk_ijepa_loss_fwd(ptr(z), ptr(h), ptr(idx), b, n, d, n_masks, idx_b, keep, dt, kind, float(eps), ptr(target),
                                        ptr(part), n_blocks, ptr(loss), stream()))
    return loss, target


def ijepa_loss_bwd(z: torch.Tensor, h: torch.Tensor, idx: torch.Tensor, kind: int, eps: float, upstream: torch.Tensor) -> torch.Tensor:
    z, h = z.contiguous(), h.contiguous()
    b, n, d = h.shape
    n_masks, idx_b, keep = idx.shape
    dz = torch.empty_like(z)
    dt = dtype_tag(z.dtype) | (dtype_tag(h.dtype) << 4)
    check(_lib.lib().mmk_ijepa_loss_bwd(ptr(z), ptr(h), ptr(idx), b, n, d, n_masks, idx_b, keep, dt, kind, float(eps), ptr(upstream),
                                        ptr(dz), stream()))
    return dz


def pred_assemble(x: torch.Tensor, pos: torch.Tensor, tok: torch.Tensor, enc_idx: torch.Tensor, pred_idx: torch.Tensor,
                  b: int, out_dtype: torch.dtype) -> torch.Tensor:
    """x [ne*b, n_ctxt, d]; pos f32 [n, d]; tok f32 [d] -> seq [np*ne*b, n_ctxt + n_pred, d]."""
    require_gpu(x)
    x = x.contiguous()
    n_enc, enc_b, n_ctxt = enc_idx.shape
    n_pm, pred_b, n_pred = pred_idx.shape
    n, d = pos.shape
    assert x.shape == (n_enc * b, n_ctxt, d)
    seq = torch.empty((n_pm * n_enc * b, n_ctxt + n_pred, d), dtype=out_dtype, device=x.device)
    dt = dtype_tag(x.dtype) | (dtype_tag(out_dtype) << 4)
    check(_lib.lib().mmk_pred_assemble(ptr(x), ptr(pos), ptr(tok), ptr(enc_idx), ptr(pred_idx), b, n, d, n_enc, n_pm, enc_b, pred_b,
                                       n_ctxt, n_pred, dt, ptr(seq), stream()))
    return seq


def pred_assemble_bwd(dseq: torch.Tensor, b: int, n_enc: int, n_pm: int, n_ctxt: int, n_pred: int, x_dtype: torch.dtype,
                      want_dx: bool, want_dtok: bool):
    dseq = dseq.contiguous()
    d = dseq.shape[-1]
    dev = dseq.device
    dx = torch.empty((n_enc * b, n_ctxt, d), dtype=x_dtype, device=dev) if want_dx else None
    dtok = part = None
    n_blocks = 0
    if want_dtok:
        n_blocks = _lib.lib().mmk_pred_tok_blocks(n_pm * n_enc * b * n_pred)
        part = torch.empty((n_blocks, d), dtype=torch.float32, device=dev)
        dtok = torch.empty(d, dtype=torch.float32, device=dev)
    dt = dtype_tag(x_dtype) | (dtype_tag(dseq.dtype) << 4)
    check(_lib.lib().mmk_pred_assemble_bwd(ptr(dseq), b, d, n_enc, n_pm, n_ctxt, n_pred, dt, ptr(dx), ptr(part), n_blocks, ptr(dtok),
                                           stream()))
    return dx, dtok


def ema_table(teacher: Sequence[torch.Tensor], student: Sequence[torch.Tensor]):
    """Device-resident pointer table for ``ema_update`` (build once, reuse every step)."""
    n = len(teacher)
    arr = (_lib.EmaEntry * n)()
    max_numel = 1
    for k, (t, s) in enumerate(zip(teacher, student)):
        require_gpu(t)
        assert t.is_contiguous() and s.is_contiguous() and t.numel() == s.numel()
        arr[k].teacher, arr[k].student, arr[k].numel = t.data_ptr(), s.data_ptr(), t.numel()
        arr[k].teacher_dtype, arr[k].student_dtype = dtype_tag(t.dtype), dtype_tag(s.dtype)
        max_numel = max(max_numel, t.numel())
    raw = bytes(arr)
    host = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
    return host.to(teacher[0].device), n, max_numel


def ema_update(table: torch.Tensor, n: int, max_numel: int, decay, true_ema: bool) -> None:
    """``decay``: a python float, or a 1-element f32 device tensor the launch reads it from (capturable)."""
    if isinstance(decay, torch.Tensor):
        assert decay.is_cuda and decay.dtype == torch.float32 and decay.numel() == 1
        check(_lib.lib().mmk_ema_update_dev(ptr(table), n, max_numel, ptr(decay), int(true_ema), stream()))
        return
    check(_lib.lib().mmk_ema_update(ptr(table), n, max_numel, float(decay), int(true_ema), stream()))


# ------------------------------------------------------------------ encoder-side row ops
def layernorm_fwd(x2: torch.Tensor, w: Optional[torch.Tensor], b: Optional[torch.Tensor], eps: float, out_dtype: torch.dtype):
    """x2 [rows, d] contiguous; w/b f32[d] or None -> (y [rows, d] of out_dtype, mean f32[rows], rstd f32[rows])."""
    require_gpu(x2)
    rows, d = x2.shape
    y = torch.empty((rows, d), dtype=out_dtype, device=x2.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x2.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x2.device)
    dt = dtype_tag(x2.dtype) | (dtype_tag(out_dtype) << 4)
    check(_lib.lib().mmk_layernorm_fwd(ptr(x2), ptr(w), ptr(b), ptr(y), ptr(mean), ptr(rstd), rows, d, float(eps), dt, stream()))
    return y, mean, rstd


def layernorm_bwd(x2: torch.Tensor, dy2: torch.Tensor, w: Optional[torch.Tensor], mean: torch.Tensor, rstd: torch.Tensor, need_wb: bool):
    rows, d = x2.shape
    dev = x2.device
    dx = torch.empty_like(x2)
    part = part2 = dw = db = None
    if need_wb:
        nb = _lib.lib().mmk_layernorm_part_blocks(rows)
        part = torch.empty((max(nb, 1), 2, d), dtype=torch.float32, device=dev)
        part2 = torch.empty((256, 2, d), dtype=torch.float32, device=dev)
        dw = torch.empty(d, dtype=torch.float32, device=dev)
        db = torch.empty(d, dtype=torch.float32, device=dev)
    dt = dtype_tag(x2.dtype) | (dtype_tag(dy2.dtype) << 4)
    check(_lib.lib().mmk_layernorm_bwd(ptr(x2), ptr(dy2), ptr(w), ptr(mean), ptr(rstd), ptr(dx), ptr(part), ptr(part2), ptr(dw), ptr(db),
                                       rows, d, dt, stream()))
    return dx, dw, db


def add_layernorm_fwd(x2: torch.Tensor, r2: torch.Tensor, w: Optional[torch.Tensor], b: Optional[torch.Tensor], eps: float,
                      out_dtype: torch.dtype, dropout_p: float = 0.0, seed: int = 0, xbias: Optional[torch.Tensor] = None,
                      twin: bool = False):
    """s = r2 + dropout(x2 + xbias), y = LN(s): x2 [rows, d] (bf16 / f32), r2 f32, xbias f32[d] or None
    -> (s f32, y of out_dtype, mean, rstd, y_twin = bf16 copy of y or None)."""
    require_gpu(x2)
    rows, d = x2.shape
    assert r2.shape == x2.shape and r2.dtype == torch.float32 and r2.is_contiguous() and x2.is_contiguous()
    s = torch.empty((rows, d), dtype=torch.float32, device=x2.device)
    y = torch.empty((rows, d), dtype=out_dtype, device=x2.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x2.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x2.device)
    y16 = torch.empty((rows, d), dtype=torch.bfloat16, device=x2.device) if twin else None
    dt = dtype_tag(x2.dtype) | (dtype_tag(out_dtype) << 4)
    check(_lib.lib().mmk_add_layernorm_fwd(ptr(x2), ptr(xbias), ptr(r2), ptr(w), ptr(b), ptr(s), ptr(y), ptr(y16), ptr(mean), ptr(rstd), rows, d,
                                           float(eps), dt, float(dropout_p), int(seed), stream()))
    return s, y, mean, rstd, y16


def add_layernorm_bwd(s2: torch.Tensor, dy2: Optional[torch.Tensor], ds_in: Optional[torch.Tensor], w: Optional[torch.Tensor], mean: torch.Tensor,
                      rstd: torch.Tensor, x_dtype: torch.dtype, need_wb: bool, dropout_p: float = 0.0, seed: int = 0, need_xbias: bool = False,
                      dy_twin: Optional[torch.Tensor] = None):
    """-> (dr f32 = ds_in + LNbwd(dy + dy_twin), dx = dropout_mask(dr) in x_dtype, dgamma, dbeta, dxbias = colsum(dx) or None)."""
    rows, d = s2.shape
    dev = s2.device
    dr = torch.empty((rows, d), dtype=torch.float32, device=dev)
    dx = torch.empty((rows, d), dtype=x_dtype, device=dev)
    part = part2 = dw = db = dxb = None
    if need_wb or need_xbias:
        npart = 3 if need_xbias else 2
        nb = _lib.lib().mmk_layernorm_part_blocks(rows)
        part = torch.empty((max(nb, 1), npart, d), dtype=torch.float32, device=dev)
        part2 = torch.empty((256, npart, d), dtype=torch.float32, device=dev)
        dw = torch.empty(d, dtype=torch.float32, device=dev)
        db = torch.empty(d, dtype=torch.float32, device=dev)
        dxb = torch.empty(d, dtype=torch.float32, device=dev) if need_xbias else None
    dt = dtype_tag(x_dtype) | (dtype_tag(torch.float32 if dy2 is None else dy2.dtype) << 4)
    check(_lib.lib().mmk_add_layernorm_bwd(ptr(s2), ptr(dy2), ptr(dy_twin), ptr(ds_in), ptr(w), ptr(mean), ptr(rstd), ptr(dr), ptr(dx), ptr(part),
                                           ptr(part2), ptr(dw), ptr(db), ptr(dxb), rows, d, dt, float(dropout_p), int(seed), stream()))
    return dr, dx, dw, db, dxb


def wgrad(dy2: torch.Tensor, x2: torch.Tensor, out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """dW [N, K] = dy2^T @ x2 for bf16 dy2 [M, N], x2 [M, K] (rows contiguous), summed in f32, returned in out_dtype."""
    require_gpu(dy2)
    assert dy2.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16 and dy2.stride(1) == 1 and x2.stride(1) == 1 and dy2.shape[0] == x2.shape[0]
    M, N = dy2.shape
    K_ = x2.shape[1]
    splits, wsf = C.c_int(0), C.c_int64(0)
    check(_lib.lib().mmk_wgrad_plan(M, N, K_, C.cast(C.pointer(splits), C.c_void_p), C.cast(C.pointer(wsf), C.c_void_p)))
    ws = torch.empty(wsf.value, dtype=torch.float32, device=dy2.device)
    dw = torch.empty((N, K_), dtype=out_dtype, device=dy2.device)
    check(_lib.lib().mmk_wgrad(ptr(dy2), ptr(x2), ptr(dw), ptr(ws), M, N, K_, dy2.stride(0), x2.stride(0), dw.stride(0), dtype_tag(out_dtype), stream()))
    return dw


def cast_transpose(w: torch.Tensor, want_plain: bool = True):
    """w [n, k] (f32 / bf16 / f16, contiguous) -> (bf16 w [n, k] or None, bf16 w^T [k, n]) in one pass over w."""
    require_gpu(w)
    assert w.dim() == 2 and w.is_contiguous()
    n, k = w.shape
    w16 = torch.empty((n, k), dtype=torch.bfloat16, device=w.device) if want_plain else None
    w16t = torch.empty((k, n), dtype=torch.bfloat16, device=w.device)
    check(_lib.lib().mmk_cast_transpose(ptr(w), ptr(w16), ptr(w16t), n, k, dtype_tag(w.dtype), stream()))
    return w16, w16t


def patchify(x: torch.Tensor, patch: int) -> torch.Tensor:
    """[B, C, H, W] (f32 / bf16 / f16, contiguous) -> bf16 [B * H/P * W/P, C * P * P]: im2col of a stride == kernel conv."""
    require_gpu(x)
    B, Cc, H, W = x.shape
    x = x.contiguous()
    out = torch.empty((B * (H // patch) * (W // patch), Cc * patch * patch), dtype=torch.bfloat16, device=x.device)
    check(_lib.lib().mmk_patchify(ptr(x), ptr(out), B, Cc, H, W, int(patch), dtype_tag(x.dtype), stream()))
    return out


def unpatchify(dcols: torch.Tensor, shape: tuple, patch: int, dtype: torch.dtype) -> torch.Tensor:
    """Backward of ``patchify`` w.r.t. the image: dcols [B * H/P * W/P, C * P * P] (bf16 / f32) -> [B, C, H, W] of ``dtype``."""
    require_gpu(dcols)
    B, Cc, H, W = shape
    dcols = dcols.contiguous()
    out = torch.empty((B, Cc, H, W), dtype=dtype, device=dcols.device)
    check(_lib.lib().mmk_unpatchify(ptr(dcols), ptr(out), B, Cc, H, W, int(patch), dtype_tag(dcols.dtype) | (dtype_tag(dtype) << 4), stream()))
    return out


def cubic_resize_rows(x: torch.Tensor, h_out: int, backward_from: Optional[int] = None) -> torch.Tensor:
    """Bicubic (align_corners) stretch of f32 ``[..., h_in, w]`` to ``[..., h_out, w]`` along the second-to-last axis; with
    ``backward_from=h_in`` the argument is the gradient w.r.t. the ``[..., h_out, w]`` output and the result the gradient w.r.t. the input."""
    require_gpu(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() >= 2 and x.shape[-1] % 4 == 0
    w = x.shape[-1]
    if backward_from is None:
        h_in = x.shape[-2]
        y = torch.empty(x.shape[:-2] + (h_out, w), dtype=x.dtype, device=x.device)
        n_img = x.numel() // (h_in * w)
        check(_lib.lib().mmk_cubic_resize_rows(ptr(x), ptr(y), n_img, h_in, int(h_out), w, 0, stream()))
        return y
    assert x.shape[-2] == h_out
    y = torch.empty(x.shape[:-2] + (backward_from, w), dtype=x.dtype, device=x.device)
    n_img = x.numel() // (h_out * w)
    check(_lib.lib().mmk_cubic_resize_rows(ptr(x), ptr(y), n_img, int(backward_from), int(h_out), w, 1, stream()))
    return y


def colsum_rows(x2: torch.Tensor) -> torch.Tensor:
    """f32 [n] column sums of a contiguous [rows, n] bf16 / f32 matrix in a fixed order (a Linear's bias gradient ``dY.sum(0)``)."""
    require_gpu(x2)
    rows, n = x2.shape
    assert x2.is_contiguous() and rows > 0 and ((x2.dtype == torch.bfloat16 and n % 8 == 0) or (x2.dtype == torch.float32 and n % 4 == 0))
    part = torch.empty((int(_lib.lib().mmk_colsum_rows_slices(rows)), n), dtype=torch.float32, device=x2.device)
    out = torch.empty(n, dtype=torch.float32, device=x2.device)
    check(_lib.lib().mmk_colsum_rows(ptr(x2), rows, n, dtype_tag(x2.dtype), ptr(part), ptr(out), stream()))
    return out


EMBEDDING_SORT_MIN_ROWS = 4096   # from this many token rows the embedding backward sorts the ids first (module seam for the tests)


def embedding_bwd(dout2: torch.Tensor, ids: torch.Tensor, vocab: int) -> torch.Tensor:
    """dW f32 [vocab, d] with dW[ids[r]] += dout2[r] (dout2 [rows, d] f32 / bf16, ids int64 [rows])."""
    require_gpu(dout2)
    rows, d = dout2.shape
    dout2, ids = dout2.contiguous(), ids.contiguous().view(-1)
    assert ids.dtype == torch.int64 and ids.numel() == rows
    dw = torch.zeros((vocab, d), dtype=torch.float32, device=dout2.device)
    if rows >= EMBEDDING_SORT_MIN_ROWS:
        # many rows: sort the ids, sum runs of the sorted order, recurse on the chunk-boundary partials (csrc/encoder_ops.hip): hot ids
        # (token-type tables, frequent tokens) never meet in same-address atomics
        sid, perm = torch.sort(ids)
        scratch = torch.empty(int(_lib.lib().mmk_embedding_bwd_scratch_bytes(rows, d)), dtype=torch.uint8, device=dout2.device)
        check(_lib.lib().mmk_embedding_bwd_sorted(ptr(dout2), ptr(sid), ptr(perm), ptr(dw), ptr(scratch), rows, d, int(vocab),
                                                  dtype_tag(dout2.dtype), stream()))
        return dw
    check(_lib.lib().mmk_embedding_bwd(ptr(dout2), ptr(ids), ptr(dw), rows, d, int(vocab), dtype_tag(dout2.dtype), stream()))
    return dw


def recall_ranks(xn: torch.Tensor, yn: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
    """xn [n, d], yn [m, d] f32 L2-normalised, pos int64[n] -> int32[n]: database rows ranked before each query's positive."""
    require_gpu(xn)
    assert xn.dtype == torch.float32 and yn.dtype == torch.float32 and pos.dtype == torch.int64
    xn, yn, pos = xn.contiguous(), yn.contiguous(), pos.contiguous()
    n, d = xn.shape
    m = yn.shape[0]
    assert yn.shape[1] == d and pos.numel() == n
    tpos = torch.empty(n, dtype=torch.float32, device=xn.device)
    rank = torch.empty(n, dtype=torch.int32, device=xn.device)
    check(_lib.lib().mmk_recall_ranks(ptr(xn), ptr(yn), ptr(pos), ptr(tpos), ptr(rank), n, m, d, stream()))
    return rank


ACT_QUICK_GELU, ACT_GELU = 0, 1


def bias_act_fwd(x2: torch.Tensor, bias: torch.Tensor, act: int) -> torch.Tensor:
    """y = act(x2 + bias): x2 [rows, d] (the bias-free Linear output), bias f32[d]."""
    require_gpu(x2)
    rows, d = x2.shape
    y = torch.empty_like(x2)
    check(_lib.lib().mmk_bias_act_fwd(ptr(x2), ptr(bias), ptr(y), rows, d, int(act), dtype_tag(x2.dtype), stream()))
    return y


def bias_act_bwd(x2: torch.Tensor, bias: torch.Tensor, dy2: torch.Tensor, act: int):
    """-> (dx = act'(x2 + bias) * dy2, dbias f32[d] = column sums of dx)."""
    rows, d = x2.shape
    dev = x2.device
    dx = torch.empty_like(x2)
    nb = _lib.lib().mmk_bias_act_part_blocks(rows)
    part = torch.empty((max(nb, 1), d), dtype=torch.float32, device=dev)
    part2 = torch.empty((256, d), dtype=torch.float32, device=dev)
    dbias = torch.empty(d, dtype=torch.float32, device=dev)
    check(_lib.lib().mmk_bias_act_bwd(ptr(x2), ptr(bias), ptr(dy2), ptr(dx), ptr(part), ptr(part2), ptr(dbias), rows, d, int(act),
                                      dtype_tag(x2.dtype), stream()))
    return dx, dbias


def mlp_gemm_supported(M: int, N: int, K: int, lda: int, ldb: int, ldc: int) -> bool:
    """True when ``csrc/mlp_gemm.hip`` serves ``[M, K] x [N, K]^T`` (M, N multiples of 256, K of 64, strides of 8)."""
    return bool(_lib.lib().mmk_mlp_gemm_supported(M, N, K, lda, ldb, ldc))


def _mlp_gemm_check(a2: torch.Tensor, b2: torch.Tensor) -> None:
    require_gpu(a2)
    if not (a2.dtype == torch.bfloat16 and b2.dtype == torch.bfloat16 and a2.dim() == 2 and b2.dim() == 2 and a2.stride(1) == 1
            and b2.stride(1) == 1 and a2.shape[1] == b2.shape[1]):
        raise ValueError("mlp_gemm: operands must be 2-D bf16 with contiguous rows and equal inner dimension")


def mlp_gemm_plain(a2: torch.Tensor, b2: torch.Tensor) -> torch.Tensor:
    """``a2 [M, K] @ b2 [N, K]^T`` in bf16 (f32 accumulation) on the MLP GEMM's main loop alone (A/B timings, parity tests)."""
    _mlp_gemm_check(a2, b2)
    M, K_ = a2.shape
    c = torch.empty((M, b2.shape[0]), dtype=torch.bfloat16, device=a2.device)
    check(_lib.lib().mmk_mlp_gemm_plain(ptr(a2), ptr(b2), ptr(c), M, b2.shape[0], K_, a2.stride(0), b2.stride(0), c.stride(0), stream()))
    return c


def mlp_gemm_fwd_act(x2: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, act: int, want_pre: bool = True):
    """-> (``act(x2 @ w^T + bias)``, bias-free pre-activation ``x2 @ w^T`` or None), both bf16: fc1 of an MLP in one kernel."""
    _mlp_gemm_check(x2, w)
    M, K_ = x2.shape
    N = w.shape[0]
    assert bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == N
    h = torch.empty((M, N), dtype=torch.bfloat16, device=x2.device)
    pre = torch.empty((M, N), dtype=torch.bfloat16, device=x2.device) if want_pre else None
    check(_lib.lib().mmk_mlp_gemm_fwd_act(ptr(x2), ptr(w), ptr(bias), ptr(h), ptr(pre), M, N, K_, x2.stride(0), w.stride(0), h.stride(0),
                                          int(act), stream()))
    return h, pre


def mlp_gemm_bwd_dact(dy2: torch.Tensor, wt: torch.Tensor, pre: torch.Tensor, bias: torch.Tensor, act: int, want_dbias: bool = True):
    """-> (``dPre = (dy2 @ wt^T) * act'(pre + bias)`` bf16, ``dbias`` f32[N] = column sums of dPre or None): the dX GEMM of fc2
    with the activation's backward in its epilogue.  ``wt`` = fc2.weight^T as [hidden, out] (K-contiguous)."""
    _mlp_gemm_check(dy2, wt)
    M, K_ = dy2.shape
    N = wt.shape[0]
    assert pre.dtype == torch.bfloat16 and pre.shape == (M, N) and pre.stride(1) == 1
    assert bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == N
    dev = dy2.device
    dpre = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    part = dbias = None
    if want_dbias:
        part = torch.empty((_lib.lib().mmk_mlp_gemm_part_rows(M), N), dtype=torch.float32, device=dev)
    check(_lib.lib().mmk_mlp_gemm_bwd_dact(ptr(dy2), ptr(wt), ptr(pre), ptr(bias), ptr(dpre), ptr(part), M, N, K_, dy2.stride(0),
                                           wt.stride(0), pre.stride(0), dpre.stride(0), int(act), stream()))
    if want_dbias:
        part2 = torch.empty((256, N), dtype=torch.float32, device=dev)
        dbias = torch.empty(N, dtype=torch.float32, device=dev)
        check(_lib.lib().mmk_colsum_f32(ptr(part), part.shape[0], N, ptr(part2), ptr(dbias), stream()))
    return dpre, dbias


def mlp_gemm_fwd_act_grad(x2: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, act: int):
    """-> (``act(x2 @ w^T + bias)``, ``act'(x2 @ w^T + bias)``), both bf16: fc1 of an MLP in one kernel that leaves behind what the
    backward multiplies by (``mlp_gemm_bwd_mul``) instead of the pre-activation."""
    _mlp_gemm_check(x2, w)
    M, K_ = x2.shape
    N = w.shape[0]
    assert bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == N
    h = torch.empty((M, N), dtype=torch.bfloat16, device=x2.device)
    g = torch.empty((M, N), dtype=torch.bfloat16, device=x2.device)
    check(_lib.lib().mmk_mlp_gemm_fwd_act_grad(ptr(x2), ptr(w), ptr(bias), ptr(h), ptr(g), M, N, K_, x2.stride(0), w.stride(0), h.stride(0),
                                               int(act), stream()))
    return h, g


def mlp_gemm_bwd_mul(dy2: torch.Tensor, wt: torch.Tensor, g: torch.Tensor, want_dbias: bool = True):
    """-> (``dPre = (dy2 @ wt^T) * g`` bf16, ``dbias`` f32[N] = column sums of dPre or None): fc2's dX GEMM with the activation's
    backward as a multiply in its epilogue; ``g`` = act'(pre + bias) from ``mlp_gemm_fwd_act_grad``, ``wt`` = fc2.weight^T."""
    _mlp_gemm_check(dy2, wt)
    M, K_ = dy2.shape
    N = wt.shape[0]
    assert g.dtype == torch.bfloat16 and g.shape == (M, N) and g.stride(1) == 1
    dev = dy2.device
    dpre = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    part = dbias = None
    if want_dbias:
        part = torch.empty((_lib.lib().mmk_mlp_gemm_part_rows(M), N), dtype=torch.float32, device=dev)
    check(_lib.lib().mmk_mlp_gemm_bwd_mul(ptr(dy2), ptr(wt), ptr(g), ptr(dpre), ptr(part), M, N, K_, dy2.stride(0), wt.stride(0),
                                          g.stride(0), dpre.stride(0), stream()))
    if want_dbias:
        part2 = torch.empty((256, N), dtype=torch.float32, device=dev)
        dbias = torch.empty(N, dtype=torch.float32, device=dev)
        check(_lib.lib().mmk_colsum_f32(ptr(part), part.shape[0], N, ptr(part2), ptr(dbias), stream()))
    return dpre, dbias


def quick_gelu_fwd(x: torch.Tensor) -> torch.Tensor:
    require_gpu(x)
    y = torch.empty_like(x)
    check(_lib.lib().mmk_quick_gelu_fwd(ptr(x), ptr(y), x.numel(), dtype_tag(x.dtype), stream()))
    return y


def quick_gelu_bwd(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    dx = torch.empty_like(x)
    check(_lib.lib().mmk_quick_gelu_bwd(ptr(x), ptr(dy), ptr(dx), x.numel(), dtype_tag(x.dtype), stream()))
    return dx


def _strides3(t: torch.Tensor):
    sb, sh, sl, sd = t.stride()
    assert sd == 1, "last dimension must be contiguous"
    return (C.c_int64 * 3)(sb, sh, sl)


def win_attn_supported(tokens: int, head_dim: int, channels: int) -> bool:
    return bool(_lib.lib().mmk_win_attn_supported(int(tokens), int(head_dim), int(channels)))


def _win_geometry(rows_total: int, n_win: int, grid, shift):
    """-> (B, img_h, img_w, shift) for ``rows_total`` token rows: window mode (grid None: rows = B * n_win * 64) or token-map mode."""
    assert rows_total % (n_win * 64) == 0, (rows_total, n_win)
    if grid is None:
        return rows_total // (n_win * 64), 0, 0, 0
    gh, gw = grid
    assert gh % 8 == 0 and gw % 8 == 0 and (gh // 8) * (gw // 8) == n_win and 0 <= shift < 8, (grid, n_win, shift)
    return rows_total // (gh * gw), gh, gw, int(shift)


def _win_operands(q, k, v, heads):
    """q / k / v: bf16 [..., C] with equal shapes and strides, last dim contiguous, rows ``ld`` elements apart with ld = C (separate
    contiguous tensors) or 3 C (the thirds of one packed [..., 3 C] projection output) -> (rows, C, dh, ld)."""
    require_gpu(q)
    Cc = q.shape[-1]
    assert q.dtype == torch.bfloat16 and k.dtype == q.dtype and v.dtype == q.dtype, "windowed attention runs on bf16 projections"
    assert k.shape == q.shape and v.shape == q.shape and q.stride() == k.stride() == v.stride() and q.stride(-1) == 1
    ld = q.stride(-2)
    rows = q.numel() // Cc
    assert ld in (Cc, 3 * Cc) and all(q.stride(i) == q.stride(i + 1) * q.shape[i + 1] for i in range(q.dim() - 2)), "rows must be evenly spaced"
    assert Cc % heads == 0 and win_attn_supported(64, Cc // heads, Cc), "windowed attention: 64-token windows, head dim 24 or 32"
    return rows, Cc, Cc // heads, ld


def _win_table(table, heads, n_win):
    assert table.dtype == torch.float32 and table.is_contiguous() and table.shape in ((1, heads, 64, 64), (n_win, heads, 64, 64)), table.shape


def win_attn_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, table: torch.Tensor, heads: int, n_win: int, scale: float,
                 grid: Optional[Tuple[int, int]] = None, shift: int = 0):
    """Windowed attention (csrc/window_attention.hip): q/k/v bf16 [B * n_win, 64, heads * dh] (window ``b * n_win + w``), table f32
    [1 | n_win, heads, 64, 64] = relative-position bias (+ mask of window position w) -> (o like q, contiguous, lse2 f32 [B * n_win, heads, 64]).
    With ``grid = (h, w)`` the tensors are ``[B, h * w, C]`` token maps instead and window ``w`` of the map rolled by ``-shift`` is gathered
    / scattered by the kernel (HF's ``roll -> window_partition -> ... -> window_reverse -> roll`` without the four copies).
    q / k / v may be the three thirds of one packed ``[..., 3 C]`` tensor (row stride 3 C)."""
    rows, Cc, dh, ld = _win_operands(q, k, v, heads)
    _win_table(table, heads, n_win)
    B, gh, gw, shift = _win_geometry(rows, n_win, grid, shift)
    o = torch.empty(q.shape, dtype=q.dtype, device=q.device)
    lse2 = torch.empty((rows // 64, heads, 64), dtype=torch.float32, device=q.device)
    check(_lib.lib().mmk_win_attn_fwd(ptr(q), ptr(k), ptr(v), ptr(table), ptr(o), ptr(lse2), B, n_win, table.shape[0], heads, dh, float(scale),
                                      gh, gw, shift, ld, stream()))
    return o, lse2


def win_attn_bwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, dout: torch.Tensor, lse2: torch.Tensor, table: torch.Tensor, heads: int,
                 n_win: int, scale: float, grid: Optional[Tuple[int, int]] = None, shift: int = 0):
    """-> (dq, dk, dv, dtable f32 [heads, 64, 64] = the gradient of the relative-position bias: dS summed over windows and batch).
    dq / dk / dv have q's layout: three contiguous tensors, or -- for packed q / k / v -- the thirds of ONE new ``[..., 3 C]`` tensor
    (``dq._base`` is it).  The kernel leaves one bias-gradient partial per workgroup; they are added here per head in a fixed order."""
    rows, Cc, dh, ld = _win_operands(q, k, v, heads)
    _win_table(table, heads, n_win)
    B, gh, gw, shift = _win_geometry(rows, n_win, grid, shift)
    assert dout.shape == q.shape and dout.dtype == q.dtype and dout.is_contiguous() and lse2.is_contiguous()
    if ld == Cc:
        dq, dk, dv = torch.empty_like(dout), torch.empty_like(dout), torch.empty_like(dout)
    else:
        dqkv = torch.empty(q.shape[:-1] + (3 * Cc,), dtype=q.dtype, device=q.device)
        dq, dk, dv = dqkv[..., :Cc], dqkv[..., Cc:2 * Cc], dqkv[..., 2 * Cc:]
    nblk = int(_lib.lib().mmk_win_attn_blocks(B, n_win, heads))
    part = torch.empty((nblk, 64 * 64), dtype=torch.float32, device=q.device)
    check(_lib.lib().mmk_win_attn_bwd(ptr(q), ptr(k), ptr(v), ptr(dout), ptr(lse2), ptr(table), ptr(dq), ptr(dk), ptr(dv), ptr(part), B, n_win,
                                      table.shape[0], heads, dh, float(scale), gh, gw, shift, ld, stream()))
    npairs = nblk // heads
    if npairs % 8 == 0:   # block id = ((group * heads + head) * 8 + x): see wa_decode_block
        dtab = part.view(npairs // 8, heads, 8, 64 * 64).sum(dim=(0, 2))
    else:                 # block id = pair * heads + head
        dtab = part.view(npairs, heads, 64 * 64).sum(0)
    return dq, dk, dv, dtab.view(heads, 64, 64)


def cls_attn_supported(L: int, dh: int) -> bool:
    return bool(_lib.lib().mmk_cls_attn_supported(int(L), int(dh)))


def _cls_kv(kv: torch.Tensor):
    """``kv``: bf16 ``[B, L, 2, H, 64]`` contiguous (the packed key | value projection) -> (k view, v view, B, L, H, batch stride, row stride)."""
    require_gpu(kv)
    assert kv.dim() == 5 and kv.shape[2] == 2 and kv.shape[4] == 64 and kv.dtype == torch.bfloat16 and kv.is_contiguous()
    B, L, _, H, dh = kv.shape
    return kv[:, :, 0], kv[:, :, 1], B, L, H, L * 2 * H * dh, 2 * H * dh


_KEYMASK_KINDS = {torch.bool: 1, torch.uint8: 1, torch.int32: 2, torch.int64: 3}   # include/mmlearn_hip.h MMK_KEYMASK_*


def attn_key_bias(mask: torch.Tensor, L: Optional[int] = None, additive: bool = False) -> torch.Tensor:
    """Key-padding mask of a batch -> the f32 ``[B, 256]`` key bias records ``attn_fwd`` / ``attn_bwd`` / ``cls_attn_*`` take
    (csrc/attention.hip ``mmk_attn_key_bias``: 0 attended, -1e30 masked, -inf beyond L).  ``mask``: ``[B, L]`` with nonzero = attended
    (bool / uint8 / int32 / int64 / float32; the tokenizer's ``attention_mask``), or with ``additive`` a float32 / bfloat16 additive mask
    (0 / ``finfo.min``); or, with ``L`` given, int32 ``[B]`` valid lengths (right padding).  Rows may be strided, the last dim not."""
    require_gpu(mask)
    if mask.dim() == 1:
        assert L is not None and mask.dtype == torch.int32 and mask.is_contiguous(), "lengths: contiguous int32 [B] plus L"
        B, kind, sb = mask.shape[0], 0, 0
    else:
        assert mask.dim() == 2 and (L is None or L == mask.shape[1]) and mask.stride(1) == 1 and mask.stride(0) >= mask.shape[1], (mask.shape, mask.stride())
        B, L = mask.shape
        sb = mask.stride(0)
        if additive:
            kind = {torch.float32: 5, torch.bfloat16: 6}[mask.dtype]
        else:
            kind = 4 if mask.dtype == torch.float32 else _KEYMASK_KINDS[mask.dtype]
    assert 1 <= L <= 256, L
    rec = torch.empty((B, 256), dtype=torch.float32, device=mask.device)
    check(_lib.lib().mmk_attn_key_bias(ptr(mask), kind, B, int(L), sb, ptr(rec), stream()))
    return rec


def _key_bias_ok(key_bias, B):
    assert key_bias is None or (key_bias.shape == (B, 256) and key_bias.dtype == torch.float32 and key_bias.is_contiguous()
                                and key_bias.is_cuda), "key_bias: the f32 [B, 256] records of attn_key_bias"


def cls_attn_fwd(q: torch.Tensor, kv: torch.Tensor, scale: float, dropout_p: float = 0.0, seed: int = 0, key_bias: Optional[torch.Tensor] = None):
    """One query per (sample, head): ``q`` bf16 ``[B, H, 64]`` contiguous, ``kv`` bf16 ``[B, L <= 256, 2, H, 64]`` ->
    (o bf16 ``[B, H, 64]``, lse2 f32 ``[B, H]``).  csrc/cls_attention.hip.  ``key_bias``: ``attn_key_bias`` records (key padding)."""
    k, v, B, L, H, sb, sl = _cls_kv(kv)
    assert q.shape == (B, H, 64) and q.dtype == torch.bfloat16 and q.is_contiguous()
    _key_bias_ok(key_bias, B)
    o = torch.empty_like(q)
    lse2 = torch.empty((B, H), dtype=torch.float32, device=q.device)
    check(_lib.lib().mmk_cls_attn_fwd(ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse2), B, H, L, 64, sb, sl, float(scale), float(dropout_p), int(seed),
                                      ptr(key_bias), stream()))
    return o, lse2


def cls_attn_bwd(q: torch.Tensor, kv: torch.Tensor, dout: torch.Tensor, lse2: torch.Tensor, scale: float, dropout_p: float = 0.0, seed: int = 0,
                 key_bias: Optional[torch.Tensor] = None):
    """-> (dq bf16 ``[B, H, 64]``, dkv bf16 ``[B, L, 2, H, 64]``: every key / value row written)."""
    k, v, B, L, H, sb, sl = _cls_kv(kv)
    assert dout.shape == q.shape and dout.dtype == torch.bfloat16 and dout.is_contiguous() and lse2.is_contiguous()
    _key_bias_ok(key_bias, B)
    dq = torch.empty_like(q)
    dkv = torch.empty_like(kv)
    check(_lib.lib().mmk_cls_attn_bwd(ptr(q), ptr(k), ptr(v), ptr(dout), ptr(lse2), ptr(dq), ptr(dkv[:, :, 0]), ptr(dkv[:, :, 1]), B, H, L, 64,
                                      sb, sl, sb, sl, float(scale), float(dropout_p), int(seed), ptr(key_bias), stream()))
    return dq, dkv


def attn_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float, dropout_p: float = 0.0, seed: int = 0,
             key_bias: Optional[torch.Tensor] = None, causal: bool = False):
    """q/k/v: [B, H, L, 64] bf16 views (last dim contiguous) -> (out [B, L, H, 64] contiguous, lse f32 [B, H, L]).
    ``dropout_p`` > 0 drops attention probabilities with the counter-based mask of ``seed`` (see csrc/attention.hip).
    ``key_bias``: the ``attn_key_bias`` records of a key-padding mask; ``causal`` (needs ``key_bias``) masks key j > query i too."""
    require_gpu(q)
    B, H, L, dh = q.shape
    _key_bias_ok(key_bias, B)
    out = torch.empty((B, L, H, dh), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, H, L), dtype=torch.float32, device=q.device)
    qs, ks, vs = _strides3(q), _strides3(k), _strides3(v)
    check(_lib.lib().mmk_attn_fwd(ptr(q), ptr(k), ptr(v), ptr(out), ptr(lse), B, H, L, dh, C.cast(qs, C.c_void_p), C.cast(ks, C.c_void_p),
                                  C.cast(vs, C.c_void_p), float(scale), float(dropout_p), int(seed), ptr(key_bias), int(bool(causal)), stream()))
    return out, lse


def attn_bwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, lse: torch.Tensor, dout: torch.Tensor, scale: float,
             dropout_p: float = 0.0, seed: int = 0, packed: bool = False, colsum: bool = False, key_bias: Optional[torch.Tensor] = None,
             causal: bool = False):
    """Backward of ``attn_fwd``: out / dout are [B, L, H, 64] contiguous.  Returns dq, dk, dv as [B, H, L, 64] VIEWS of
    [B, L, H, 64] buffers (the layout the q/k/v projections' backward consumes without a copy), or with ``packed`` one
    [B, L, 3, H, 64] buffer holding the three (the gradient of a fused QKV projection's output); ``packed`` + ``colsum``
    returns ``(dqkv, f32 [3 * H * 64])``, the second being ``dqkv.view(-1, 3 * H * 64).sum(0)`` -- the projection's bias
    gradient -- accumulated from per-tile sums the kernel takes while it stores, instead of a pass over the gradient."""
    require_gpu(q)
    B, H, L, dh = q.shape
    _key_bias_ok(key_bias, B)
    assert out.is_contiguous() and dout.is_contiguous() and out.shape == (B, L, H, dh) == dout.shape
    if packed:
        dqkv = torch.empty((B, L, 3, H, dh), dtype=q.dtype, device=q.device)
        dq, dk, dv = dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2]
    else:
        dq, dk, dv = (torch.empty((B, L, H, dh), dtype=q.dtype, device=q.device) for _ in range(3))
    gs = (C.c_int64 * 2)(dq.stride(0), dq.stride(1))
    delta = torch.empty((B * H, 2, 256), dtype=torch.float32, device=q.device)  # lse2 / delta records (csrc/attention.hip)
    qs, ks, vs = _strides3(q), _strides3(k), _strides3(v)
    in_kernel = packed and colsum and bool(_lib.lib().mmk_attn_bwd_has_colsum(L))   # else: one pass over the gradient afterwards
    part = torch.empty((B * ((L + 31) // 32), 3 * H * dh), dtype=torch.float32, device=q.device) if in_kernel else None
    check(_lib.lib().mmk_attn_bwd(ptr(q), ptr(k), ptr(v), ptr(out), ptr(dout), ptr(lse), ptr(delta), ptr(dq), ptr(dk), ptr(dv), B, H, L, dh,
                                  C.cast(qs, C.c_void_p), C.cast(ks, C.c_void_p), C.cast(vs, C.c_void_p), C.cast(gs, C.c_void_p),
                                  float(scale), float(dropout_p), int(seed), ptr(part), ptr(key_bias), int(bool(causal)), stream()))
    if packed and colsum:
        if not in_kernel:
            return dqkv, dqkv.view(B * L, -1).sum(0, dtype=torch.float32)
        n = part.shape[1]                       # the per-tile partials, summed in the two fixed-order stages of the row kernels
        part2 = torch.empty((256, n), dtype=torch.float32, device=q.device)
        dbias = torch.empty(n, dtype=torch.float32, device=q.device)
        check(_lib.lib().mmk_colsum_f32(ptr(part), part.shape[0], n, ptr(part2), ptr(dbias), stream()))
        return dqkv, dbias
    if packed:
        return dqkv
    return dq.transpose(1, 2), dk.transpose(1, 2), dv.transpose(1, 2)
