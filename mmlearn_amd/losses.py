"""Drop-in ``ContrastiveLoss`` for mmlearn, computed by hand-written HIP kernels on MI355X.

Boundary (mmlearn/modules/losses/contrastive.py:40-47,59-65; docs/user_guide.md:149-165):

    loss = ContrastiveLoss(l2_normalize, local_loss, gather_with_grad, modality_alignment, cache_labels)
    loss(embeddings, example_ids, logit_scale, modality_loss_pairs) -> 0-dim tensor

with ``embeddings`` keyed by ``Modality.embedding`` ("rgb_embedding"), ``example_ids`` keyed by
modality name with int64 ``[B, 2]`` values, ``logit_scale`` a 0-dim tensor and
``modality_loss_pairs`` a list of ``LossPairSpec(modalities, weight)``.

What runs where
---------------
* device matcher (``find_matching_indices``), gather+normalise+cast packing, the similarity tiles
  with fused row log-sum-exp, the gradient tiles and the ``G @ Y`` products are HIP kernels
  (``mmlearn_amd/csrc``), reached through ``mmlearn_amd.kernels``;
* across ranks the [N, N] matrix is sharded by rows: two all-gathers
  (embeddings, ids) replace the reference's >= 12 collectives, then every rank computes only its
  own row blocks and one all-reduce of the per-row LSEs (+ loss partial sums) makes the full
  gradient local.  The four ``(local_loss, gather_with_grad)`` cells reproduce the reference's
  per-rank loss values and gradient scalings.

There is no eager/CPU fallback: CPU tensors raise ``RuntimeError``.
"""

from __future__ import annotations

import contextlib

import math
from dataclasses import dataclass, field
from typing import Any, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist
from torch import nn

from . import kernels as K
from ._lib import COMPUTE_BF16, COMPUTE_F32
from .modalities import Modalities
from .registry import store

_ACCUM_DTYPE = torch.float32


@dataclass
class LossPairSpec:
    """Specification for a pair of modalities to compute the contrastive loss
    (mmlearn/tasks/contrastive_pretraining.py:44-52)."""

    modalities: tuple[str, str]
    weight: float = 1.0


def find_matching_indices(first_example_ids: torch.Tensor, second_example_ids: torch.Tensor):
    """Device version of mmlearn.datasets.core.find_matching_indices (example.py:101-166): the
    indices of matching examples of the first and second tensor, in ``torch.where`` order."""
    if not isinstance(first_example_ids, torch.Tensor) or not isinstance(second_example_ids, torch.Tensor):
        raise TypeError(f"Expected inputs to be tensors, but got {type(first_example_ids)} and {type(second_example_ids)}.")
    for name, t in (("first_example_ids", first_example_ids), ("second_example_ids", second_example_ids)):
        if not (t.ndim == 2 and t.shape[1] == 2):
            raise ValueError(f"Expected argument `{name}` to be a tensor of shape (N, 2), but got shape {t.shape}.")
    m = K.match_ids(first_example_ids.to(torch.int64), second_example_ids.to(torch.int64))
    if m.identity:
        ar = torch.arange(m.n, device=first_example_ids.device)
        return ar, ar.clone()
    return m.idx_a.to(torch.int64), m.idx_b.to(torch.int64)


def _unpack_keys(keys: torch.Tensor) -> torch.Tensor:
    """int64 [N] keys ``dataset_index << 32 | example_index`` (mmlearn_amd.wire.pack_example_ids) -> [N, 2]."""
    keys = keys.to(torch.int64)
    return torch.stack([(keys >> 32) & 0xFFFFFFFF, keys & 0xFFFFFFFF], dim=1)


# ----------------------------------------------------------------------------------------------
_FUSED_MAX_GROUPS = 4   # launches of the one-launch kernel per loss call before the tiled path is the better deal


@dataclass
class _View:
    """One modality as seen by the loss: rows of every rank (compact, rank order)."""

    name: str
    local: Optional[torch.Tensor]        # this rank's [B_r, D] embedding (None if the rank lacks the modality)
    src: torch.Tensor                    # matrix holding all ranks' rows
    ids: torch.Tensor                    # int64 [N_m, 2]
    rows: Optional[torch.Tensor]         # int32 [N_m]: compact index -> row of src (None = identity)
    counts: list[int]                    # rows per rank

    def my_range(self, rank: int) -> tuple[int, int]:
        lo = sum(self.counts[:rank])
        return lo, lo + self.counts[rank]


@dataclass
class _Pair:
    spec: Any
    dirs: list = field(default_factory=list)       # K.Direction, at most [a->b, b->a]
    roles: list = field(default_factory=list)      # "a" / "b" per direction
    r_global: int = 0
    kappa_loss: float = 0.0                        # factor of (sum_a + sum_b) in the loss value
    own: dict = field(default_factory=dict)        # role -> (p0, p1) contiguous or int64 index tensor
    exch_off: int = 0                              # offset of this pair's block in the exchange buffer
    col_perm: dict = field(default_factory=dict)   # role -> int64 tensor (column permutation) or None
    lse_adj: dict = field(default_factory=dict)    # role -> tensor subtracted from lse_col (local+gwg, uneven)
    loss_terms: list = field(default_factory=list)


def _compose(rows: Optional[torch.Tensor], idx: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if rows is None:
        return idx
    if idx is None:
        return rows
    return rows[idx.long()].contiguous()


def _slice_rows(t: torch.Tensor, p0: int) -> torch.Tensor:
    if not p0:
        return t
    slicer = getattr(K, "slice_packed", None)   # keeps the row norms attached to a packed operand
    return slicer(t, p0) if slicer is not None else t[p0:]


def _all_gather(t: torch.Tensor, world: int) -> torch.Tensor:
    """One all-gather of a contiguous tensor -> [world, *t.shape] (flat buffers: valid for RCCL and gloo)."""
    flat = t.contiguous().view(-1)
    out = torch.empty(world * flat.numel(), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, flat)
    return out.view(world, *t.shape)


class _Run:
    """One evaluation of the loss: forward state kept for the backward pass."""

    def __init__(self, owner: "ContrastiveLoss", embeddings, example_ids, logit_scale, pairs, fully_paired=False):
        self.o = owner
        self.embeddings = embeddings
        # ids arrive as [B, 2] (dataset_index, example_index) or as packed int64 [B] keys (mmlearn_amd.wire)
        self.example_ids = {k: (_unpack_keys(v) if isinstance(v, torch.Tensor) and v.dim() == 1 else v) for k, v in example_ids.items()}
        self.paired_hint = bool(fully_paired)   # this rank's batch: every modality carries the same id column
        self.paired = False                     # agreed by all ranks (set in build_views)
        self.logit_scale = logit_scale
        self.pair_specs = pairs
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.pairs: list[_Pair] = []
        self.inputs: list[torch.Tensor] = []
        self.input_names: list[str] = []
        # gradients for EVERY input, whatever its requires_grad says: inside a compiled graph (compiled.py) the operator receives plain
        # tensors and autograd's bookkeeping lives in the traced graph
        self.all_grads = False

    # ------------------------------------------------------------------ set-up
    def compute_mode(self) -> int:
        dts = {t.dtype for t in self.embeddings.values()}
        if len(dts) > 1:
            raise ValueError(f"all embeddings must share one dtype, got {sorted(map(str, dts))}")
        dt = dts.pop()
        if self.o.compute_dtype is not None:
            return COMPUTE_BF16 if self.o.compute_dtype == torch.bfloat16 else COMPUTE_F32
        if dt == torch.bfloat16:
            return COMPUTE_BF16
        if dt == torch.float32 and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16:
            return COMPUTE_BF16  # the reference's matmul runs in bf16 under Lightning's bf16-mixed
        return COMPUTE_F32  # fp32, and fp16 (torchmetrics _safe_matmul up-casts fp16 to fp32)

    def build_views(self) -> dict[str, _View]:
        emb = {k: v for k, v in self.embeddings.items()}
        key_to_name = {}
        for key in emb:
            name = key[: -len("_embedding")] if key.endswith("_embedding") else key
            if Modalities.has_modality(name):
                name = Modalities.get_modality(name).name
            key_to_name[key] = name
        local = {key_to_name[k]: v for k, v in emb.items()}
        for name, t in local.items():
            if t.ndim != 2:
                raise ValueError(f"embedding of modality {name!r} must be [B, D], got {tuple(t.shape)}")
            if name not in self.example_ids:
                raise KeyError(f"example_ids has no entry for modality {name!r}")
        if self.world == 1 and not (self.o._force_gather and dist.is_available() and dist.is_initialized()):
            self.paired = self.paired_hint
            return {n: _View(n, t, t.detach().contiguous(), self.example_ids[n].to(torch.int64), None, [t.shape[0]])
                    for n, t in local.items()}
        return self._gather_views(local)

    def _gather_views(self, local: dict[str, torch.Tensor]) -> dict[str, _View]:
        """Two all-gathers (embedding rows, id rows) instead of the reference's barrier + object gather +
        per-key shape gather + per-key data gather (contrastive.py:431-578)."""
        names_all = [m.name for m in Modalities.list_modalities()]
        for n in local:
            if n not in names_all:
                raise ValueError(f"modality {n!r} is not registered")
        pre = self.o._take_prefetched(local)
        if pre is not None:  # gathers were started right after each encoder (overlapped with the next one)
            self.o.prefetched_gathers_used += 1
            views = {}
            cur = torch.cuda.current_stream() if next(iter(local.values())).is_cuda else None
            for n, (all_e, all_i, works) in pre.items():
                for w in works:
                    w.wait()  # stream-level wait on the collective's stream, no host block
                if cur is not None:   # the buffers may have been allocated on a tower's side stream
                    all_e.record_stream(cur)
                    all_i.record_stream(cur)
                views[n] = _View(n, local[n], all_e, all_i, None, [local[n].shape[0]] * self.world)
            return views
        any_t = next(iter(local.values()))
        dev, dt, d = any_t.device, any_t.dtype, any_t.shape[1]
        W, rank = self.world, self.rank
        if self.o.static_shapes:
            self.o._check_static_shapes({n: t.shape[0] for n, t in local.items()})
            counts = {n: [t.shape[0]] * W for n, t in local.items()}
        else:
            header = torch.tensor([local[n].shape[0] if n in local else -1 for n in names_all] + [int(self.paired_hint), d],
                                  dtype=torch.int64, device=dev)
            table = _all_gather(header, W).tolist()
            if any(row[-1] != d for row in table):
                raise ValueError("embedding dimension differs across ranks")
            # the pairing flag rides in the header that is exchanged anyway: identity pairing needs it from every rank
            # (and from ranks that hold the same modalities: a rank without one side of a pair has nothing to pair)
            present = [tuple(c >= 0 for c in row[:-2]) for row in table]
            self.paired = all(row[-2] == 1 for row in table) and all(p == present[0] for p in present)
            counts = {n: [max(table[r][i], 0) for r in range(W)] for i, n in enumerate(names_all)
                      if any(table[r][i] >= 0 for r in range(W))}
        names = sorted(counts)  # the reference iterates the sorted key union (contrastive.py:466)
        bmax = {n: max(counts[n]) for n in names}
        roff, tot = {}, 0
        for n in names:
            roff[n] = tot
            tot += bmax[n]
        send_e = torch.zeros((tot, d), dtype=dt, device=dev)
        send_i = None if self.paired else torch.zeros((tot, 2), dtype=torch.int64, device=dev)
        for n, t in local.items():
            send_e[roff[n]: roff[n] + t.shape[0]].copy_(t.detach())
            if send_i is not None:
                send_i[roff[n]: roff[n] + t.shape[0]].copy_(self.example_ids[n])
        all_e = _all_gather(send_e, W).view(W * tot, d)
        all_i = None if self.paired else _all_gather(send_i, W).view(W * tot, 2)   # paired everywhere: ids are not needed
        views = {}
        for n in names:
            rows_np = np.concatenate([np.arange(counts[n][r], dtype=np.int32) + (r * tot + roff[n]) for r in range(W)]) \
                if sum(counts[n]) else np.zeros(0, np.int32)
            rows = torch.from_numpy(rows_np).to(dev)
            ids = None if all_i is None else (all_i[rows.long()] if rows.numel() else all_i[:0]).contiguous()
            views[n] = _View(n, local.get(n), all_e, ids, rows, counts[n])
        return views

    @staticmethod
    def _paired_match(counts_a, counts_b, ma: str, mb: str) -> "K.Match":
        """Identity pairing promised by the batch's ``fully_paired`` flag: pair p is (p, p); no kernel, no read-back."""
        if list(counts_a) != list(counts_b):
            raise ValueError(f"fully_paired batch, but modalities {ma!r} and {mb!r} have different row counts "
                             f"({list(counts_a)} vs {list(counts_b)})")
        return K.Match(int(sum(counts_a)), True, None, None)

    # ------------------------------------------------------------------ forward
    def forward(self) -> Optional[torch.Tensor]:
        o = self.o
        scale = self.logit_scale
        K.require_gpu(scale, "logit_scale")
        self.compute = self.compute_mode()
        # needs_grad is decided by the caller (autograd.Function.forward runs with grad mode off)
        self.scale32 = scale.detach().to(torch.float32).reshape(1).contiguous()
        views = self.build_views()
        self.views = views
        W, rank = self.world, self.rank
        local_mode = o.local_loss and W > 1
        self.local_mode = local_mode
        self.d = next(iter(self.embeddings.values())).shape[1]
        dev = scale.device

        # ---- matching per pair
        local_counts_needed = []
        for spec in self.pair_specs:
            ma, mb = (Modalities.get_modality(m).name if Modalities.has_modality(m) else m for m in spec.modalities)
            if ma not in views or mb not in views:
                continue  # contrastive.py:266-274 / :303-307
            va, vb = views[ma], views[mb]
            mg = self._paired_match(va.counts, vb.counts, ma, mb) if self.paired else o._matched(va.ids, vb.ids, (ma, mb, "global"))
            if mg.n == 0:
                continue  # :283-287 / :314-316
            p = _Pair(spec=spec, r_global=mg.n)
            p.ma, p.mb, p.mg = ma, mb, mg
            self.pairs.append(p)

        # ---- small batches on one rank: the whole loss AND its gradients in one resident-grid launch (csrc/clip_fused.hip)
        self.fused = None
        if self.pairs and self._try_fused(views):
            # not kept on self: the returned tensor's grad_fn owns this run, a reference back would make a cycle, and the run's
            # workspace goes back to the pool when the run is dropped (by reference count, not by a later gc pass)
            loss, self.fused_loss = self.fused_loss, None
            return loss

        # ---- operand packing per pair
        n_dirs = min(2 * len(self.pairs), K.MAX_DIRS_PER_CALL)
        for p in self.pairs:
            ma, mb, mg = p.ma, p.mb, p.mg
            va, vb = views[ma], views[mb]
            # The transposed copies feed the dX = G Y GEMM of the two-launch backward.  At W > 1 every direction is a row shard
            # (this rank's rows against all gathered columns), which at CLIP widths runs as one kernel that reads Y^T out of the
            # row-major tile it already holds (csrc/clip_bwd.hip): no copy then.  Decided on this rank's row counts (side a's
            # transpose serves the direction whose rows are side b's, and the other way round; a side without rows here has no
            # direction); should the directions come out different in the end, clip_backward makes the copy it needs.
            def wants_t(own_rows_other_side: int) -> bool:
                if not self.needs_grad:
                    return False
                if W == 1:   # one rank: the pair's two directions, each over all mg.n rows and columns
                    return not K.pair_runs_untied(mg.n, va.src.shape[1], self.compute, n_dirs)
                return own_rows_other_side > 0 and not K.backward_recomputes_on_chip(own_rows_other_side, mg.n, va.src.shape[1], self.compute, n_dirs)

            (p.a_g, p.a_gt), (p.b_g, p.b_gt) = K.pack_rows_many(
                [(va.src, _compose(va.rows, mg.idx_a), mg.n, o.l2_normalize, wants_t(vb.counts[rank])),
                 (vb.src, _compose(vb.rows, mg.idx_b), mg.n, o.l2_normalize, wants_t(va.counts[rank]))], self.compute)
            if local_mode:
                has_local = va.local is not None and vb.local is not None
                if not has_local:
                    p.ml = K.Match(0, False, None, None)
                elif self.paired:
                    p.ml = self._paired_match([va.local.shape[0]], [vb.local.shape[0]], ma, mb)
                else:
                    p.ml = o._matched(self.example_ids[ma].to(torch.int64), self.example_ids[mb].to(torch.int64), (ma, mb, "local"))
                local_counts_needed.append(p)

        if local_mode and self.pairs:
            # per-rank row counts of every pair (the reference all-gathers them per pair, contrastive.py:196-206).  When the
            # pairing of the GATHERED ids is the identity (every rank sees the same answer, so the branch is taken
            # collectively) rank r's local pairing is the identity over its own rows: the counts are known, no exchange,
            # no host read.
            if all(p.mg.identity and list(views[p.ma].counts) == list(views[p.mb].counts) for p in self.pairs):
                for p in self.pairs:
                    p.local_sizes = list(views[p.ma].counts)
            else:
                mine = torch.tensor([p.ml.n for p in self.pairs], dtype=torch.int64, device=dev)
                table = _all_gather(mine, W).tolist()
                for k, p in enumerate(self.pairs):
                    p.local_sizes = [table[r][k] for r in range(W)]

        dirs_all = []
        for p in self.pairs:
            self._build_dirs(p)
            dirs_all += p.dirs
        self.fused_loss = None
        if dirs_all:
            # one rank, plain pairs: the reduction launch can form the loss value itself (weights = kappa of each direction)
            w = [p.kappa_loss for p in self.pairs for _ in p.dirs] if W == 1 else None
            self.fused_loss = K.clip_forward(dirs_all, self.d, self.compute, self.scale32, loss_weights=w)
        self.align_dirs = []
        if o.modality_alignment:
            self._build_alignment(views, dev)

        if not self.pairs and not o.modality_alignment:
            return None
        loss = self._finish_forward(dev) if self.pairs else None
        if o.modality_alignment:
            # (1/M) * sum over ALL rows of (pos_r/npos_r + neg_r/nneg_r); every rank holds its own rows' share
            if self.align_dirs:
                al = K.reduce_sums([d.loss_part for d in self.align_dirs], [1.0 / self.align_m] * len(self.align_dirs))
            else:
                al = torch.zeros((), dtype=torch.float32, device=dev)
            if W > 1:
                dist.all_reduce(al)
            loss = al if loss is None else loss + al
        return loss

    def _try_fused(self, views: dict) -> bool:
        """One rank, no alignment term, no in-loss normalisation, bf16 arithmetic, pairs of <= 1024 matched rows: ONE launch per
        group of pairs whose tile grid is co-resident (usually one group) computes the loss value and leaves the raw gradient
        sums in a pooled workspace (``kernels.clip_fused_forward``); ``backward`` is one more launch per group.  Anything else
        takes the tiled multi-launch path."""
        o = self.o
        fused_plan = getattr(K, "clip_fused_plan", None)   # absent from the CPU test double
        if (fused_plan is None or self.world != 1 or o.modality_alignment or o.l2_normalize or self.compute != COMPUTE_BF16
                or any(v.rows is not None for v in views.values())):
            return False
        src = {m: views[m].src for p in self.pairs for m in (p.ma, p.mb)}
        # f32 rows that carry their own rounding to bf16 (``ops.l2_normalize`` under bf16 autocast): read the copy -- the kernel
        # would round every row to the same bits while staging it
        def _twin(t):   # (copy, version of t when the copy was written): stale after an in-place edit of t
            pair = getattr(t, "_mmk_bf16_nograd", None)
            return pair[0] if pair is not None and not t.is_inference() and pair[1] == t._version else None

        twins = {m: _twin(views[m].local) for m in src}
        if all(t is not None and t.dtype == torch.bfloat16 and t.shape == src[m].shape and t.device == src[m].device and t.is_contiguous()
               and src[m].dtype == torch.float32 for m, t in twins.items()):
            src = twins
        first = next(iter(src.values()))
        if any(t.dtype != first.dtype or t.shape[1] != self.d for t in src.values()):
            return False
        # Pairs in launch groups: as many consecutive pairs per launch as the kernel takes and the device holds at once (three pairs
        # of 1024 rows are 768 tiles for 512 resident slots: two launches, 2 + 1).  A pair that does not fit a launch of its own,
        # or more than _FUSED_MAX_GROUPS launches, sends everything to the tiled path.
        groups, cur, cur_plan = [], [], None
        for p in self.pairs:
            plan = fused_plan(first.device, [q.mg.n for q in cur] + [p.mg.n], self.d, first.dtype) if len(cur) < K.FUSED_MAX_PAIRS else None
            if plan is None:
                if cur:
                    groups.append((cur, cur_plan))
                cur, cur_plan = [p], fused_plan(first.device, [p.mg.n], self.d, first.dtype)
                if cur_plan is None:
                    return False
            else:
                cur, cur_plan = cur + [p], plan
        groups.append((cur, cur_plan))
        if len(groups) > _FUSED_MAX_GROUPS:
            return False
        runs, loss = [], None
        for members, plan in groups:
            part, run = K.clip_fused_forward(plan, [(src[p.ma], src[p.mb], p.mg.idx_a, p.mg.idx_b, p.mg.n, float(p.spec.weight))
                                                    for p in members], self.d, self.scale32, self.needs_grad)
            runs.append((members, run))
            loss = part if loss is None else loss + part
        self.fused, self.fused_loss = runs, loss
        return True

    def _fused_backward(self, grad_out: torch.Tensor):
        dev = self.scale32.device
        upstream = grad_out.detach().to(torch.float32).reshape(1)
        key_of, name_of = {}, {}
        for key in self.embeddings:
            name = key[: -len("_embedding")] if key.endswith("_embedding") else key
            name = Modalities.get_modality(name).name if Modalities.has_modality(name) else name
            key_of[name], name_of[key] = key, name
        writers: dict[str, int] = {}
        for p in self.pairs:
            for n in (p.ma, p.mb):
                writers[n] = writers.get(n, 0) + 1
        acc, covered = {}, {}
        for p in self.pairs:
            for n, rep in ((p.ma, p.mg.repeats_a), (p.mb, p.mg.repeats_b)):
                acc[n] = acc.get(n, False) or writers[n] > 1 or (rep and not p.mg.identity)
                covered[n] = p.mg.identity and p.mg.n == self.embeddings[key_of[n]].shape[0]   # every row written exactly once
        mixed = len({self.embeddings[key_of[n]].dtype for n in acc}) > 1 or (any(acc.values()) and not all(acc.values()))
        grads = {}
        for n, a_ in acc.items():
            t = self.embeddings[key_of[n]]
            dt = _ACCUM_DTYPE if (a_ or mixed) else t.dtype
            grads[n] = (torch.empty if (covered[n] and not a_) else torch.zeros)(t.shape, dtype=dt, device=dev)
        runs = self.fused   # kept: the kernels' raw sums are only read here, a second backward (retain_graph) repeats the launches
        want_ds = self.logit_scale.requires_grad or self.all_grads
        # d loss / d scale accumulates into a word the (first) forward launch left at zero; a repeated backward gets a fresh one
        # (the first one's may have become the parameter's .grad)
        head = runs[0][1]
        ds_acc = head.ds_acc if head.n_bwd == 0 else torch.zeros(1, dtype=torch.float32, device=dev)
        head.n_bwd += 1
        for members, run in runs:
            K.clip_fused_backward(run, [(grads[p.ma], grads[p.mb], acc[p.ma], acc[p.mb]) for p in members], self.scale32, upstream,
                                  ds_acc if want_ds else None)
        out = []
        for key, t in self.embeddings.items():
            g = grads.get(name_of[key])
            if g is None:
                out.append(torch.zeros_like(t) if (t.requires_grad or self.all_grads) else None)
            else:
                out.append(g if g.dtype == t.dtype else g.to(t.dtype))
        ds = ds_acc.reshape(self.logit_scale.shape).to(self.logit_scale.dtype) if want_ds else None
        return ds, out

    def _build_alignment(self, views: dict, dev) -> None:
        """Rows of the modality-alignment BCE (contrastive.py:344-413) owned by this rank: one problem per modality.
        The concatenation follows the reference's dict order (insertion order at W = 1, sorted keys after a gather)
        and the positives use its non-cumulative block offsets (quirk Q2)."""
        o, W, rank = self.o, self.world, self.rank
        order = list(views) if W == 1 else sorted(views, key=lambda n: Modalities.get_modality(n).embedding
                                                  if Modalities.has_modality(n) else n)
        sizes = [sum(views[n].counts) for n in order]
        m_total = sum(sizes)
        self.align_m = m_total
        self.align_order = order
        hmax = np.arange(1, m_total + 1, dtype=np.int32)
        for k, n in enumerate(sizes):
            off = 0 if k == 0 else sizes[k - 1]
            hmax[off: off + n] = np.maximum(hmax[off: off + n], off + n)
        hmax_t = torch.from_numpy(hmax).to(dev)
        if all(views[n].rows is None for n in order):  # one compact matrix per modality (W = 1, or prefetched gathers)
            src = torch.cat([views[n].src for n in order], 0)
            idx = None
        else:  # one packed gather holding every modality: select the rows
            src = views[order[0]].src
            idx = torch.cat([views[n].rows for n in order]).contiguous()
        f_all, f_all_t = K.pack_rows(src, idx, m_total, o.l2_normalize, self.compute, self.needs_grad)
        # gradient multiplier of the gathered shards: own shard re-inserted x1, dist_nn gather xW,
        # no gradient path in the (local_loss, no gather_with_grad) cell
        mult = 1.0 if W == 1 else (float(W) if o.gather_with_grad else (0.0 if o.local_loss else 1.0))
        off = 0
        for n, size in zip(order, sizes):
            v = views[n]
            if v.local is not None and v.local.shape[0] > 0:
                lo, hi = v.my_range(rank)
                dr = K.Direction(x=_slice_rows(f_all, off + lo), y=f_all, y_t=f_all_t, r=hi - lo, c=m_total, label_off=off + lo,
                                 kappa=mult / m_total, ds_kappa=1.0 / m_total, mode=1, hmax=hmax_t)
                dr.modality = n
                self.align_dirs.append(dr)
            off += size
        if self.align_dirs:
            K.clip_forward(self.align_dirs, self.d, self.compute, self.scale32)

    def _own_rows(self, view: _View, idx: Optional[torch.Tensor], n: int):
        """Rows p of the matched list whose `view` row belongs to this rank -> (contiguous?, p0, p1 | index array)."""
        lo, hi = view.my_range(self.rank)
        if self.world == 1:
            return True, 0, n, None
        if idx is None:  # identity pairing: p == compact index
            p0, p1 = min(lo, n), min(hi, n)
            return True, p0, p1, None
        host = idx.cpu().numpy()
        sel = np.nonzero((host >= lo) & (host < hi))[0]
        if sel.size == 0:
            return True, 0, 0, None
        if sel[-1] - sel[0] + 1 == sel.size:
            return True, int(sel[0]), int(sel[-1]) + 1, None
        return False, 0, 0, sel

    def _build_dirs(self, p: _Pair) -> None:
        o, W, rank = self.o, self.world, self.rank
        va, vb = self.views[p.ma], self.views[p.mb]
        mg, R = p.mg, p.r_global
        w = float(p.spec.weight)
        dev = p.a_g.device
        if not self.local_mode:
            kg = w / (2.0 * R)
            p.kappa_loss = kg
            for role, view, idx, xg, yg, ygt in (("a", va, mg.idx_a, p.a_g, p.b_g, p.b_gt), ("b", vb, mg.idx_b, p.b_g, p.a_g, p.a_gt)):
                contiguous, p0, p1, sel = self._own_rows(view, idx, R)
                lo, _ = view.my_range(rank)
                if contiguous:
                    r = p1 - p0
                    p.own[role] = (p0, p1)
                    if r == 0:
                        continue
                    x, y, yt, label_off = _slice_rows(xg, p0), yg, ygt, p0
                    p.col_perm[role] = None
                    dx_rows = None if idx is None else ((idx[p0:p1] - lo).to(torch.int32) if W > 1 else idx[p0:p1])
                else:  # owned rows scattered over the matched list: put their partners first in the column order
                    r = sel.size
                    rest = np.setdiff1d(np.arange(R), sel, assume_unique=True)
                    perm = torch.from_numpy(np.concatenate([sel, rest])).to(dev)
                    sel_t = torch.from_numpy(sel).to(dev)
                    p.own[role] = sel_t
                    p.col_perm[role] = perm
                    other_view, other_idx = (vb, mg.idx_b) if role == "a" else (va, mg.idx_a)
                    x, _ = K.pack_rows(view.src, _compose(view.rows, idx[sel_t].contiguous()), r, o.l2_normalize, self.compute, False)
                    y, yt = K.pack_rows(other_view.src, _compose(other_view.rows, other_idx[perm].contiguous()), R, o.l2_normalize, self.compute,
                                        self.needs_grad and not K.backward_recomputes_on_chip(r, R, view.src.shape[1], self.compute, 2))
                    label_off = 0
                    dx_rows = (idx[sel_t] - lo).to(torch.int32).contiguous()
                dr = K.Direction(x=x, y=y, y_t=yt, r=r, c=R, label_off=label_off,
                                 kappa=kg * (W if (o.gather_with_grad and W > 1) else 1.0), ds_kappa=kg)
                if role == "b":
                    dr.s_row = dr.s_col = dr.s_diag = 0.0  # the b->a tiles are the transposed a->b tiles
                dr.dx_rows = dx_rows
                p.dirs.append(dr)
                p.roles.append(role)
        else:
            ml = p.ml
            sizes = p.local_sizes
            offs = [0]
            for s in sizes:
                offs.append(offs[-1] + s)
            p.offs = offs
            Rl = ml.n
            p.kappa_loss = w / (2.0 * Rl) if Rl else 0.0
            if Rl == 0:
                return
            kl = w / (2.0 * Rl)
            reuse = ml.identity and mg.identity and va.my_range(rank)[0] == offs[rank] == vb.my_range(rank)[0] and offs[rank] + Rl <= R
            for role, view, lidx, xg, yg, ygt in (("a", va, ml.idx_a, p.a_g, p.b_g, p.b_gt), ("b", vb, ml.idx_b, p.b_g, p.a_g, p.a_gt)):
                if reuse:
                    x = _slice_rows(xg, offs[rank])
                else:
                    x, _ = K.pack_rows(view.local, lidx, Rl, o.l2_normalize, self.compute, False)
                dr = K.Direction(x=x, y=yg, y_t=ygt, r=Rl, c=R, label_off=offs[rank], kappa=kl, ds_kappa=kl)
                dr.s_row, dr.s_col, dr.s_diag = 1.0, 0.0, 1.0
                if o.gather_with_grad:
                    dr.c_row, dr.c_col, dr.c_diag = 1.0, 1.0, 2.0
                    if any(s != Rl for s in sizes if s):  # uneven means: fold kappa_r'/kappa_r into the column LSEs
                        ratio = torch.cat([torch.full((s,), math.log(Rl / s) if s else 0.0, dtype=torch.float32) for s in sizes]).to(dev)
                        p.lse_adj[role] = ratio
                else:
                    dr.c_row, dr.c_col, dr.c_diag = 1.0, 0.0, 1.0
                dr.dx_rows = lidx
                p.own[role] = (offs[rank], offs[rank] + Rl)
                p.col_perm[role] = None
                p.dirs.append(dr)
                p.roles.append(role)

    def _finish_forward(self, dev) -> torch.Tensor:
        o, W = self.o, self.world
        exchange_lse = W > 1 and self.needs_grad and (not self.local_mode or o.gather_with_grad)
        exchange_sum = W > 1 and not self.local_mode
        terms, weights = [], []
        if exchange_lse or exchange_sum:
            size = 0
            for p in self.pairs:
                p.exch_off = size
                size += 2 * p.r_global + 2
            buf = torch.zeros(size, dtype=torch.float32, device=dev)
            for p in self.pairs:
                R = p.r_global
                for dr, role in zip(p.dirs, p.roles):
                    base = p.exch_off + (0 if role == "a" else R)
                    own = p.own[role]
                    if isinstance(own, tuple):
                        buf[base + own[0]: base + own[1]].copy_(dr.lse)
                    else:
                        buf[base: base + R].index_copy_(0, own, dr.lse)
                    k = p.exch_off + 2 * R + (0 if role == "a" else 1)
                    K.reduce_sums([dr.loss_part], [1.0], separate=True, out=buf[k:k + 1])
            dist.all_reduce(buf)
            self.exch = buf
            for p in self.pairs:
                R = p.r_global
                p.lse_global = {"a": buf[p.exch_off: p.exch_off + R], "b": buf[p.exch_off + R: p.exch_off + 2 * R]}
                if exchange_sum:
                    terms += [buf[p.exch_off + 2 * R: p.exch_off + 2 * R + 1], buf[p.exch_off + 2 * R + 1: p.exch_off + 2 * R + 2]]
                    weights += [p.kappa_loss, p.kappa_loss]
        if not exchange_sum:
            for p in self.pairs:
                for dr in p.dirs:
                    terms.append(dr.loss_part)
                    weights.append(p.kappa_loss)
        if W == 1:
            for p in self.pairs:
                by_role = dict(zip(p.roles, p.dirs))
                p.lse_global = {r: d.lse for r, d in by_role.items()}
        if not terms:  # this rank owns no rows (e.g. it lacks a modality): graph-attached zero
            return torch.zeros((), dtype=torch.float32, device=dev)
        if self.fused_loss is not None and not exchange_sum:
            loss, self.fused_loss = self.fused_loss, None   # (no reference from the run back to its own output: see forward)
            return loss
        out = None
        for i0 in range(0, len(terms), 2 * K.MAX_DIRS_PER_CALL):
            part = K.reduce_sums(terms[i0:i0 + 2 * K.MAX_DIRS_PER_CALL], weights[i0:i0 + 2 * K.MAX_DIRS_PER_CALL])
            out = part if out is None else out + part
        return out

    # ------------------------------------------------------------------ backward
    def backward(self, grad_out: torch.Tensor):
        if getattr(self, "fused", None) is not None:
            return self._fused_backward(grad_out)
        o, W = self.o, self.world
        dev = self.scale32.device
        upstream = grad_out.detach().to(torch.float32).reshape(1).contiguous()
        key_of = {}
        for key in self.embeddings:
            name = key[: -len("_embedding")] if key.endswith("_embedding") else key
            name = Modalities.get_modality(name).name if Modalities.has_modality(name) else name
            key_of[name] = key
        # how many directions write each modality's gradient
        writers: dict[str, int] = {}
        for p in self.pairs:
            for role in p.roles:
                n = p.ma if role == "a" else p.mb
                writers[n] = writers.get(n, 0) + 1
        for dr in self.align_dirs:
            writers[dr.modality] = writers.get(dr.modality, 0) + 1
        grads: dict[str, torch.Tensor] = {}
        accumulate: dict[str, bool] = {}
        for dr in self.align_dirs:
            accumulate[dr.modality] = writers[dr.modality] > 1
        for p in self.pairs:
            for role in p.roles:
                n = p.ma if role == "a" else p.mb
                m = p.ml if self.local_mode else p.mg
                rep = (m.repeats_a if role == "a" else m.repeats_b) if not m.identity else False
                accumulate[n] = accumulate.get(n, False) or rep or writers[n] > 1
        # one launch takes one gradient dtype: when some modality needs the f32 accumulating scatter (repeated rows, several
        # writers) and the embeddings are not f32, every buffer of the call is f32 and is cast once at the end
        mixed = any(accumulate.values()) and any(self.embeddings[key_of[n]].dtype != _ACCUM_DTYPE for n in accumulate)
        for n, acc in accumulate.items():
            t = self.embeddings[key_of[n]]
            grads[n] = torch.zeros(t.shape, dtype=_ACCUM_DTYPE if (acc or mixed) else t.dtype, device=dev)
        dirs_all = []
        for p in self.pairs:
            for dr, role in zip(p.dirs, p.roles):
                n = p.ma if role == "a" else p.mb
                other = "b" if role == "a" else "a"
                if dr.c_col != 0.0 or dr.s_col != 0.0:
                    lc = p.lse_global[other]
                    if role in p.lse_adj:
                        lc = lc - p.lse_adj[role]
                    if p.col_perm.get(role) is not None:
                        lc = lc[p.col_perm[role]]
                    dr.lse_col = lc.contiguous()
                dr.dx = grads[n]
                dr.dx_accumulate = accumulate[n]
                if o.l2_normalize:
                    dr.normalize = True
                    dr.src = self.embeddings[key_of[n]].detach().contiguous()
                dirs_all.append(dr)
        for dr in self.align_dirs:
            dr.dx = grads[dr.modality]
            dr.dx_accumulate = accumulate[dr.modality]
            if o.l2_normalize:
                dr.normalize = True
                dr.src = self.embeddings[key_of[dr.modality]].detach().contiguous()
        dscale = torch.zeros(1, dtype=torch.float32, device=dev)
        dscale_align = torch.zeros(1, dtype=torch.float32, device=dev) if self.align_dirs or o.modality_alignment else None
        every = dirs_all + self.align_dirs
        if every:
            deferred_norm = o.l2_normalize and any(d.dx_accumulate for d in every)
            if deferred_norm:
                for d_ in every:
                    d_.normalize = False
            if self.align_dirs:
                K.clip_backward(self.align_dirs, self.d, self.compute, self.scale32, upstream, dscale_align)
        if dirs_all:
            if o.l2_normalize and any(d.dx_accumulate for d in every):
                # the normalise-backward of an accumulating scatter needs the summed upstream gradient first:
                # run un-normalised, then apply the projection once per row.
                for d_ in dirs_all:
                    d_.normalize = False
                K.clip_backward(dirs_all, self.d, self.compute, self.scale32, upstream, dscale)
                for n in grads:
                    src = self.embeddings[key_of[n]].detach()
                    g = grads[n]
                    y, inv = K.l2norm_fwd(src.float())
                    # rows were packed normalised, so G@Y is d/dy; map to d/dx
                    grads[n] = K.l2norm_bwd(src.float(), g.float(), inv)
            else:
                K.clip_backward(dirs_all, self.d, self.compute, self.scale32, upstream, dscale)
        elif every and o.l2_normalize and any(d.dx_accumulate for d in every):
            for n in grads:  # alignment only: apply the deferred normalise-backward
                src = self.embeddings[key_of[n]].detach()
                _, inv = K.l2norm_fwd(src.float())
                grads[n] = K.l2norm_bwd(src.float(), grads[n].float(), inv)
        if W > 1 and not self.local_mode:
            dist.all_reduce(dscale)  # every rank returns the full d loss / d scale (reference: identical graphs)
        if dscale_align is not None:
            if W > 1:
                dist.all_reduce(dscale_align)  # each rank summed its own rows of the (replicated) alignment term
            dscale = dscale + dscale_align
        out = []
        for key, t in self.embeddings.items():
            name = next(n for n, k in key_of.items() if k == key)
            g = grads.get(name)
            if g is None:
                out.append(torch.zeros_like(t) if (t.requires_grad or self.all_grads) else None)
            else:
                out.append(g if g.dtype == t.dtype else g.to(t.dtype))
        ds = dscale.reshape(self.logit_scale.shape).to(self.logit_scale.dtype) if (self.logit_scale.requires_grad or self.all_grads) else None
        return ds, out


class _ContrastiveFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, run: _Run, logit_scale: torch.Tensor, *embs: torch.Tensor):
        loss = run.forward()
        ctx.run = run
        if loss is None:  # no pair matched; the caller returns the reference's graph-less constant
            return torch.zeros((), device=logit_scale.device)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        ds, gs = ctx.run.backward(grad_out)
        return (None, ds, *gs)


@store(group="modules/losses", name="ContrastiveLossHIP")
class ContrastiveLoss(nn.Module):
    """Contrastive (CLIP-style InfoNCE) loss on MI355X; same constructor and call signature as
    ``mmlearn.modules.losses.ContrastiveLoss`` (contrastive.py:19-57).

    Parameters
    ----------
    l2_normalize, local_loss, gather_with_grad, modality_alignment, cache_labels
        As in the reference.  ``cache_labels`` is accepted and ignored (labels are never
        materialised: label(i) = offset + i inside the kernels).  ``modality_alignment=True`` adds the
        reference's BCE term over the concatenated modalities, including its non-cumulative positive
        offsets for the 3rd+ modality.
    compute_dtype : torch.dtype, optional
        Force the arithmetic of the similarity products (``torch.bfloat16`` -> bf16 MFMA,
        ``torch.float32`` -> exact-f32 MFMA).  Default: bf16 for bf16 inputs or under bf16 autocast,
        f32 otherwise.
    static_shapes : bool
        Multi-rank only: promise that every rank holds the same modalities with the same batch
        size, which removes the per-step header exchange (one collective + host sync).  Checked on the
        first step and whenever this rank's shapes change (see ``_check_static_shapes`` for the one case
        the check cannot see: samplers that shorten the batch on some ranks only).
    """

    def __init__(self, l2_normalize: bool = False, local_loss: bool = False, gather_with_grad: bool = False,
                 modality_alignment: bool = False, cache_labels: bool = False, compute_dtype: Optional[torch.dtype] = None,
                 static_shapes: bool = False):
        super().__init__()
        if compute_dtype not in (None, torch.bfloat16, torch.float32):
            raise ValueError("compute_dtype must be None, torch.bfloat16 or torch.float32")
        self.l2_normalize = l2_normalize
        self.local_loss = local_loss
        self.gather_with_grad = gather_with_grad
        self.modality_alignment = modality_alignment
        self.cache_labels = cache_labels
        self.compute_dtype = compute_dtype
        self.static_shapes = static_shapes
        from . import compiled

        self._site_id = compiled.register_site(self)   # how a torch.compile'd graph names this module (compiled.py)
        self._pending: dict[str, tuple] = {}
        self.prefetched_gathers_used = 0   # forward() calls that consumed gathers started by prefetch_gather
        self.prefetched_matches_used = 0   # pairings answered by prefetch_match
        self._force_gather = False   # test seam: run the gather path (packed all-gathers, prefetch) on a 1-rank process group
        self._pending_match: list = []
        self._early_ids: dict[str, tuple] = {}
        self._match_stream = None
        self._static_validated: dict[str, int] = {}
        # graph capture with the id matcher in the graph: the pair COUNT (and the identity / repeated-row flags) are host values
        # that size the loss's launches, so a captured step takes them from the last eager step with the same id-column shapes
        # and checks on the device that every replayed batch still has them (capture_mismatch, NaN loss otherwise)
        self._match_seen: dict[tuple, tuple] = {}
        self._capture_poison: Optional[torch.Tensor] = None
        self.capture_mismatch: Optional[torch.Tensor] = None   # device bool: some replayed batch matched differently than captured

    @contextlib.contextmanager
    def forcing_gather(self, on: bool = True):
        """Test seam as a context manager: run the gather path (packed all-gathers, prefetch) on a 1-rank process group inside the
        block, and put the previous setting back afterwards."""
        saved, self._force_gather = self._force_gather, bool(on)
        try:
            yield self
        finally:
            self._force_gather = saved

    # ------------------------------------------------------------------ static_shapes: a checked promise
    def _check_static_shapes(self, rows: dict[str, int]) -> None:
        """``static_shapes=True`` promises that every rank holds the same modalities with the same row counts; a broken
        promise would otherwise show up as mismatched collectives (a hang or corrupted rows).  Whenever THIS rank sees a
        (modality, rows) it has not validated yet -- the first step, or a shorter last batch, which ``DistributedSampler``
        hands to every rank at the same step -- the ranks exchange a small header and compare (one collective + one host
        read, only on those steps).  LIMITATION: the trigger is this rank's own shapes, so it is only rank-symmetric when every
        rank's shapes change at the same step (what ``DistributedSampler`` and the reference's padded samplers give).  A
        sampler that hands a shorter batch to SOME ranks only makes the triggering ranks issue the header collective while the
        others go straight to the embedding gather -- mismatched collectives, i.e. the hang this check exists to prevent.  Use
        ``static_shapes=False`` with such samplers; ``MMK_CHECK_STATIC_SHAPES=1`` validates on every call (every rank always
        joins the header collective, at the price of one small collective + host read per step)."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        import os

        if all(self._static_validated.get(n) == r for n, r in rows.items()) and not os.environ.get("MMK_CHECK_STATIC_SHAPES"):
            return
        names_all = [m.name for m in Modalities.list_modalities()]
        unknown = [n for n in rows if n not in names_all]
        if unknown:
            raise ValueError(f"modality {unknown[0]!r} is not registered")
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        header = torch.tensor([rows.get(n, -1) for n in names_all], dtype=torch.int64, device=dev)
        table = _all_gather(header, dist.get_world_size()).tolist()
        if any(row != table[0] for row in table):
            bad = {names_all[i]: [row[i] for row in table] for i in range(len(names_all)) if len({row[i] for row in table}) > 1}
            raise ValueError("ContrastiveLoss(static_shapes=True): ranks disagree on the rows per modality (-1 = modality absent) "
                             f"{bad}; use static_shapes=False for ragged / missing-modality batches")
        self._static_validated.update(rows)

    # ------------------------------------------------------------------ gather / encoder overlap
    def prefetch_gather(self, modality: str, embedding: torch.Tensor, example_ids: torch.Tensor) -> None:
        """Start this modality's all-gather now (asynchronously, on the process group's own stream) so that it
        overlaps with the encoders that still have to run; ``forward`` picks the result up.  Called by
        ``ContrastivePretraining.forward`` after each ``encode``.  Only with ``static_shapes=True`` (all ranks hold
        the same modalities and batch size); otherwise a no-op and the packed gather in ``forward`` is used."""
        if not (self.static_shapes and dist.is_available() and dist.is_initialized()
                and (dist.get_world_size() > 1 or self._force_gather)):
            return
        world = dist.get_world_size()
        self._check_static_shapes({modality: embedding.shape[0]})
        e = embedding.detach().contiguous()
        all_e = torch.empty((world * e.shape[0], e.shape[1]), dtype=e.dtype, device=e.device)
        works = [dist.all_gather_into_tensor(all_e.view(-1), e.view(-1), async_op=True)]
        early = self._early_ids.get(modality)
        if early is not None and early[0] is example_ids and early[1].shape[0] == world * e.shape[0]:
            i, all_i = early[2], early[1]   # gathered (and matched) ahead of the encoders by prefetch_match
        else:
            i = example_ids.to(torch.int64).contiguous()
            all_i = torch.empty((world * i.shape[0], 2), dtype=torch.int64, device=i.device)
            works.append(dist.all_gather_into_tensor(all_i.view(-1), i.view(-1), async_op=True))
        self._pending[modality] = (embedding, all_e, all_i, works, (e, i))

    # ------------------------------------------------------------------ matcher / encoder overlap
    def prefetch_match(self, example_ids: dict[str, torch.Tensor], modality_loss_pairs: Sequence[Any],
                       stream: Optional["torch.cuda.Stream"] = None) -> None:
        """Run the id matcher NOW, on its own stream, and park its status in pinned memory: the ids exist before the
        encoders run, so by the time ``forward`` asks for the pairing the answer has been on the host for tens of
        milliseconds and reading it does not drain the compute stream (without this the read-back is the one host sync
        of the loss path and holds back the queueing of the backward pass).  Called by ``ContrastivePretraining.forward``
        before the encoders (``stream``: a side stream to run on instead of the loss's own).  Across ranks (with
        ``static_shapes=True``) the id columns are all-gathered here as well -- they do not depend on the encoders -- and
        ``prefetch_gather`` then only moves embeddings."""
        self._pending_match, self._early_ids = [], {}
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            return   # inside a graph capture there is no read-back to get ahead of: forward() runs the matcher in the graph
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        gathered = world > 1 or (self._force_gather and dist.is_available() and dist.is_initialized())
        if gathered and not self.static_shapes:
            return   # row counts are only known after the size header: the matcher stays in forward()
        names = []
        for spec in modality_loss_pairs:
            ma, mb = (Modalities.get_modality(m).name if Modalities.has_modality(m) else m for m in spec.modalities)
            ia, ib = example_ids.get(ma), example_ids.get(mb)
            if not all(isinstance(t, torch.Tensor) and t.is_cuda and t.dim() == 2 and t.dtype == torch.int64 and t.shape[0] for t in (ia, ib)):
                continue
            names.append((ma, mb))
        if not names:
            return
        if gathered:
            self._check_static_shapes({n: example_ids[n].shape[0] for pair in names for n in pair})
        if stream is None:   # the caller may lend a side stream it already has (HW queues are few: keep the stream count low)
            if self._match_stream is None:
                self._match_stream = torch.cuda.Stream()
            stream = self._match_stream
        main = torch.cuda.current_stream()
        stream.wait_stream(main)   # the ids may still be in flight (H2D copy) on the caller's stream
        with torch.cuda.stream(stream):
            cols = {}
            for n in sorted({n for pair in names for n in pair}):   # same order on every rank: these are collectives
                ids = example_ids[n]
                if gathered:
                    # the ids do not depend on the encoders: gather them now; prefetch_gather then moves embeddings only
                    src = ids.contiguous()
                    all_i = torch.empty((world * src.shape[0], 2), dtype=torch.int64, device=src.device)
                    dist.all_gather_into_tensor(all_i.view(-1), src.view(-1), async_op=True).wait()   # stream-level wait
                    self._early_ids[n] = (ids, all_i, src)
                    cols[n] = all_i
                else:
                    cols[n] = ids
            todo = [(cols[ma], cols[mb]) for ma, mb in names]
            if gathered and self.local_loss:   # local_loss also pairs this rank's own rows
                todo += [(example_ids[ma], example_ids[mb]) for ma, mb in names]
            for ia, ib in todo:
                pm = K.match_ids_launch(ia, ib, read_back_async=True)
                pm.side_stream = stream
                self._pending_match.append(pm)

    def _matched_in_capture(self, ids_a: torch.Tensor, ids_b: torch.Tensor, who: tuple = ()) -> "K.Match":
        """The matcher inside a HIP-graph capture: its kernels are captured (every replay pairs that batch's ids on the device), its
        16-byte status is NOT read back -- the counts that size the launches come from the last eager step with these shapes, and
        a device-side comparison poisons the loss (NaN) and raises ``capture_mismatch`` when a replayed batch pairs differently
        (another pair count, identity vs. permuted order, repeated ids).  A permutation-paired batch of the captured size always
        replays correctly: warm up with a representative (shuffled) batch.  The expectation is kept per modality pair (``who``), so
        pairs whose id columns have equal shapes but pair differently (three modalities, one of them shuffled) capture too."""
        key = (who, tuple(ids_a.shape), tuple(ids_b.shape))
        seen = self._match_seen.get(key)
        if seen is None:
            raise RuntimeError("mmlearn_amd.ContrastiveLoss: the id matcher cannot learn its pair count during graph capture -- run one "
                               "eager step with a batch of the same shapes first (or pass fully_paired=True)")
        total, ident, rep_a, rep_b, expect = seen
        pm = K.match_ids_launch(ids_a, ids_b)
        if total > pm.idx_a.numel():
            raise RuntimeError("mmlearn_amd.ContrastiveLoss: heavily duplicated ids need a second matcher pass; not capturable")
        bad = (pm.status != expect).any()
        self._capture_poison = bad if self._capture_poison is None else (self._capture_poison | bad)
        if ident:
            return K.Match(total, True, None, None)
        return K.Match(total, False, pm.idx_a[:total], pm.idx_b[:total], bool(rep_a), bool(rep_b))

    def _matched(self, ids_a: torch.Tensor, ids_b: torch.Tensor, who: tuple = ()) -> "K.Match":
        """The pairing of two id columns: the prefetched answer if it was computed for exactly these tensors.  ``who`` names the
        call (modality pair, global / local pairing): the key of what a captured step may expect of it."""
        if ids_a.is_cuda and torch.cuda.is_current_stream_capturing():
            return self._matched_in_capture(ids_a, ids_b, who)
        m = self._matched_eager(ids_a, ids_b)
        if ids_a.is_cuda:
            key = (who, tuple(ids_a.shape), tuple(ids_b.shape))
            seen = self._match_seen.get(key)
            if seen is None or seen[:4] != (m.n, m.identity, m.repeats_a, m.repeats_b):   # (device copy made once per change)
                expect = torch.tensor([m.n, int(m.identity), int(m.repeats_a), int(m.repeats_b)], dtype=torch.int32, device=ids_a.device)
                self._match_seen[key] = (m.n, m.identity, m.repeats_a, m.repeats_b, expect)
            if self.capture_mismatch is None:
                self.capture_mismatch = torch.zeros((), dtype=torch.bool, device=ids_a.device)
        return m

    def _matched_eager(self, ids_a: torch.Tensor, ids_b: torch.Tensor) -> "K.Match":
        for k, pm in enumerate(self._pending_match):
            if pm.ids_a.data_ptr() == ids_a.data_ptr() and pm.ids_b.data_ptr() == ids_b.data_ptr() \
                    and pm.ids_a.shape == ids_a.shape and pm.ids_b.shape == ids_b.shape:
                del self._pending_match[k]
                self.prefetched_matches_used += 1
                main = torch.cuda.current_stream()
                main.wait_stream(pm.side_stream)          # idx_a / idx_b were written on the matcher's stream
                for t in (pm.idx_a, pm.idx_b, pm.counts, pm.status):
                    t.record_stream(main)
                return K.match_ids_finish(pm)
        return K.match_ids(ids_a, ids_b)

    def _take_prefetched(self, local: dict[str, torch.Tensor]):
        """Prefetched gathers, if there is one for every local modality and it belongs to exactly these tensors."""
        pending, self._pending = self._pending, {}
        if not pending or set(pending) != set(local) or any(pending[n][0] is not local[n] for n in local):
            for _, _, _, works, _ in pending.values():
                for w in works:
                    w.wait()  # never leave a collective un-joined
            return None
        return {n: (p[1], p[2], p[3]) for n, p in pending.items()}

    def forward(self, embeddings: dict[str, torch.Tensor], example_ids: dict[str, torch.Tensor], logit_scale: torch.Tensor,
                modality_loss_pairs: Sequence[Any], fully_paired: Optional[bool] = None) -> torch.Tensor:
        """``fully_paired=True`` (``batch["fully_paired"]`` of ``mmlearn_amd.wire.DefaultDataCollator``) states that all
        modalities of THIS rank's batch carry the same id column in the same order: rows pair by position, the id
        matcher, its status read-back and -- when every rank says so in the size header -- the id all-gather are
        skipped.  With ``static_shapes=True`` across ranks there is no header to agree in and the flag is ignored."""
        if not embeddings:
            raise ValueError("embeddings is empty")
        if not isinstance(logit_scale, torch.Tensor):
            raise TypeError("logit_scale must be a 0-dim tensor")
        from . import compiled

        if compiled.is_compiling():   # torch.compile (mmlearn/cli/run.py:139): the whole call is ONE traced operator
            return compiled.contrastive_loss(self, embeddings, example_ids, logit_scale, modality_loss_pairs, fully_paired)
        for t in embeddings.values():
            K.require_gpu(t, "embedding")
        run = _Run(self, embeddings, example_ids, logit_scale, list(modality_loss_pairs), fully_paired)
        self._capture_poison = None
        try:
            loss = self._forward(run, embeddings, logit_scale)
            if self._capture_poison is not None:   # captured matcher: a replayed batch that pairs differently must not train silently
                self.capture_mismatch.logical_or_(self._capture_poison)   # persistent flag (created by the eager warm-up step)
                loss = torch.where(self._capture_poison, torch.full_like(loss, float("nan")), loss)
            return loss
        finally:
            self._pending_match, self._early_ids = [], {}   # answers belong to one batch
            self._capture_poison = None

    def __setstate__(self, state):
        super().__setstate__(state)
        from . import compiled

        self._site_id = compiled.register_site(self)   # a copy / an unpickled module is its own call site

    def _forward(self, run: "_Run", embeddings, logit_scale) -> torch.Tensor:
        first = next(iter(embeddings.values()))
        run.needs_grad = torch.is_grad_enabled() and (logit_scale.requires_grad or any(t.requires_grad for t in embeddings.values()))
        if not run.needs_grad:
            with torch.no_grad():
                loss = run.forward()
        else:
            loss = _ContrastiveFn.apply(run, logit_scale, *embeddings.values())
            if not run.pairs and not self.modality_alignment:
                loss = None
        if loss is None:
            # no loss to compute (e.g. no paired data in batch): constant zero, contrastive.py:151-158
            return torch.tensor(0.0, device=logit_scale.device, dtype=first.dtype)
        # CE runs in f32 under autocast; without it the reference's loss has the embeddings' dtype
        if first.dtype != torch.float32 and not torch.is_autocast_enabled():
            loss = loss.to(first.dtype)
        return loss
