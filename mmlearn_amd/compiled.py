"""``torch.compile`` support: the ops on the contrastive step's path as ``torch.library`` custom ops with fake (meta) implementations.

The reference plumbs whole-task compilation (mmlearn/cli/run.py:139: ``torch.compile(task, **compile_kwargs)``;
mmlearn/conf/__init__.py:140-149).  This package reaches its HIP kernels through ``ctypes``, which TorchDynamo cannot trace: without
this module it breaks the graph at every kernel call (correct, but ``fullgraph=True`` fails and nothing is captured across the
ops).  Here every kernel entry of the step is ONE opaque operator to the tracer:

=====================================  ==================================================================================
``mmlearn_amd::l2_normalize_fwd/bwd``   ``F.normalize(x, dim=-1)`` (tasks/contrastive_pretraining.py:428-429) and its backward
``mmlearn_amd::contrastive_loss_fwd``   the whole of ``ContrastiveLoss.forward`` (modules/losses/contrastive.py:59-160): gather,
                                        matching, similarity statistics, CE in both directions -- host logic included
``mmlearn_amd::contrastive_loss_bwd``   its gradients w.r.t. every embedding and the logit scale
=====================================  ==================================================================================

Each has a fake implementation (shapes and dtypes only), so Dynamo + AOT autograd trace THROUGH them: the small task compiles with
``fullgraph=True`` (``aot_eager``) and trains to bit-identical parameters (tests/test_graph_capture_gpu.py).  The loss is an operator
with Python state (which pairs matched, the packed operands, workspaces): the forward op parks its run in a small registry and
returns a token tensor that the backward op redeems; a run whose backward never comes (evaluation) is evicted by the next ones.

The towers' fused blocks (``mmlearn_amd.fused``) are not wrapped: a compiled task with accelerated HF towers still breaks the graph
at their kernels (and stays correct) -- the launch-free form of those steps is HIP-graph capture (``mmlearn_amd.graph``).
"""

from __future__ import annotations

import itertools
import weakref
from collections import OrderedDict
from typing import Any, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import kernels as K

_SITES: "weakref.WeakValueDictionary[int, Any]" = weakref.WeakValueDictionary()   # loss modules by their site id
_site_counter = itertools.count(1)
_RUNS: "OrderedDict[int, Any]" = OrderedDict()    # token -> forward run awaiting its backward
_run_counter = itertools.count(1)
MAX_PARKED_RUNS = 16


def register_site(module) -> int:
    """An integer a traced graph can carry in place of the loss module (called once per module, from its constructor)."""
    sid = next(_site_counter)
    _SITES[sid] = module
    return sid


def is_compiling() -> bool:
    fn = getattr(torch.compiler, "is_compiling", None)
    return bool(fn()) if fn is not None else False


# ------------------------------------------------------------------------------------------------ l2 normalise
@torch.library.custom_op("mmlearn_amd::l2_normalize_fwd", mutates_args=())
def _l2n_fwd(x: Tensor) -> Tuple[Tensor, Tensor]:
    y, inv = K.l2norm_fwd(x)
    return y, inv


@_l2n_fwd.register_fake
def _(x):
    rows = x.numel() // max(x.shape[-1], 1)
    return torch.empty_like(x, memory_format=torch.contiguous_format), x.new_empty((rows,), dtype=torch.float32)


@torch.library.custom_op("mmlearn_amd::l2_normalize_bwd", mutates_args=())
def _l2n_bwd(x: Tensor, dy: Tensor, inv: Tensor) -> Tensor:
    return K.l2norm_bwd(x, dy.to(x.dtype), inv)


@_l2n_bwd.register_fake
def _(x, dy, inv):
    return torch.empty_like(x, memory_format=torch.contiguous_format)


def _l2n_setup(ctx, inputs, output):
    (x,) = inputs
    ctx.save_for_backward(x, output[1])


def _l2n_backward(ctx, dy, _dinv):
    x, inv = ctx.saved_tensors
    return _l2n_bwd(x, dy, inv)


_l2n_fwd.register_autograd(_l2n_backward, setup_context=_l2n_setup)


def l2_normalize(x: Tensor) -> Tensor:
    """The traced form of ``ops.l2_normalize`` (no bf16 twin: the loss rounds the rows itself, to the same bits)."""
    if torch.is_autocast_enabled() and x.dtype != torch.float32:
        x = x.float()
    return _l2n_fwd(x)[0]


# ------------------------------------------------------------------------------------------------ contrastive loss
def _encode_meta(emb_keys: Sequence[str], id_keys: Sequence[str], pairs: Sequence[Any], fully_paired: Optional[bool]) -> str:
    """Everything of a loss call that is not a tensor, as one string operand (custom ops take tensors and scalars)."""
    ps = ",".join(f"{p.modalities[0]}:{p.modalities[1]}:{float(p.weight)!r}" for p in pairs)
    return "|".join(emb_keys) + ";" + "|".join(id_keys) + ";" + ps + ";" + ("1" if fully_paired is True else "0" if fully_paired is False else "")


def _decode_meta(meta: str):
    from .tasks.contrastive_pretraining import LossPairSpec

    ek, ik, ps, fp = meta.split(";")
    pairs = []
    for item in filter(None, ps.split(",")):
        a, b, w = item.split(":")
        pairs.append(LossPairSpec(modalities=(a, b), weight=float(w)))
    return ek.split("|"), [k for k in ik.split("|") if k], pairs, {"1": True, "0": False, "": None}[fp]


def _park(run) -> int:
    tok = next(_run_counter)
    _RUNS[tok] = run
    while len(_RUNS) > MAX_PARKED_RUNS:   # forward passes whose backward never came (evaluation under compile)
        _RUNS.popitem(last=False)
    return tok


@torch.library.custom_op("mmlearn_amd::contrastive_loss_fwd", mutates_args=())
def _loss_fwd(embs: List[Tensor], ids: List[Tensor], logit_scale: Tensor, site: int, meta: str, needs_grad: bool,
              bf16_autocast: bool) -> Tuple[Tensor, Tensor]:
    from .losses import _Run

    module = _SITES.get(site)
    if module is None:
        raise RuntimeError("mmlearn_amd::contrastive_loss_fwd: the loss module of this compiled graph no longer exists")
    emb_keys, id_keys, pairs, fully_paired = _decode_meta(meta)
    run = _Run(module, dict(zip(emb_keys, embs)), dict(zip(id_keys, ids)), logit_scale, pairs, fully_paired)
    run.needs_grad = run.all_grads = bool(needs_grad)
    try:
        # a compiled graph runs with autocast switched off (its casts are already in the graph); the loss picks its arithmetic from the
        # autocast state of the CALL, which the traced wrapper recorded
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bool(bf16_autocast)):
            loss = run.forward()
    finally:
        module._pending_match, module._early_ids = [], {}
    if loss is None:   # no pair matched (contrastive.py:151-158 returns a graph-less constant; here the zero stays attached to the graph)
        loss = torch.zeros((), dtype=torch.float32, device=logit_scale.device)
    tok = _park(run) if needs_grad else 0
    # the token is a host tensor -- nothing on the device ever reads it
    return loss.reshape(()).to(torch.float32), torch.tensor([tok], dtype=torch.int64)


@_loss_fwd.register_fake
def _(embs, ids, logit_scale, site, meta, needs_grad, bf16_autocast):
    return logit_scale.new_empty((), dtype=torch.float32), torch.empty((1,), dtype=torch.int64, device="cpu")


@torch.library.custom_op("mmlearn_amd::contrastive_loss_bwd", mutates_args=())
def _loss_bwd(token: Tensor, grad_out: Tensor, embs: List[Tensor], logit_scale: Tensor) -> Tuple[Tensor, List[Tensor]]:
    run = _RUNS.pop(int(token[0]), None)
    if run is None:
        raise RuntimeError("mmlearn_amd::contrastive_loss_bwd: no parked forward run for this token (backward called twice, or more than "
                           f"{MAX_PARKED_RUNS} forward passes without a backward in between)")
    ds, gs = run.backward(grad_out)
    ds = torch.zeros_like(logit_scale) if ds is None else ds.reshape(logit_scale.shape).to(logit_scale.dtype)
    return ds, [torch.zeros_like(e) if g is None else g for g, e in zip(gs, embs)]


@_loss_bwd.register_fake
def _(token, grad_out, embs, logit_scale):
    return torch.empty_like(logit_scale), [torch.empty_like(e, memory_format=torch.contiguous_format) for e in embs]


def _loss_setup(ctx, inputs, output):
    embs, ids, logit_scale, _site, _meta, _needs_grad, _bf16 = inputs
    ctx.save_for_backward(output[1], logit_scale, *embs)
    ctx.n_ids = len(ids)


def _loss_backward(ctx, gloss, _gtoken):
    token, logit_scale, *embs = ctx.saved_tensors
    ds, gembs = _loss_bwd(token, gloss, list(embs), logit_scale)
    return list(gembs), [None] * ctx.n_ids, ds, None, None, None, None   # same structure as the inputs: the id list gets a list


_loss_fwd.register_autograd(_loss_backward, setup_context=_loss_setup)


def contrastive_loss(module, embeddings: dict, example_ids: dict, logit_scale: Tensor, modality_loss_pairs: Sequence[Any],
                     fully_paired: Optional[bool]) -> Tensor:
    """The traced form of ``ContrastiveLoss.forward``: one operator in, one scalar out."""
    emb_keys = list(embeddings.keys())
    id_keys = list(example_ids.keys())
    embs = [embeddings[k] for k in emb_keys]
    needs_grad = torch.is_grad_enabled() and (logit_scale.requires_grad or any(t.requires_grad for t in embs))
    meta = _encode_meta(emb_keys, id_keys, list(modality_loss_pairs), fully_paired)
    bf16_autocast = torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
    loss, token = _loss_fwd(embs, [example_ids[k] for k in id_keys], logit_scale, module._site_id, meta, needs_grad, bf16_autocast)
    first = embs[0]
    if first.dtype != torch.float32 and not torch.is_autocast_enabled():
        loss = loss.to(first.dtype)
    return loss
