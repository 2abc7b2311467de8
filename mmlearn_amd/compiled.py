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
``mmlearn_amd::tower_fwd/bwd``          an encoder patched by ``accelerate_encoder`` (tasks/contrastive_pretraining.py:400-431 ``encode``):
                                        its whole forward, and the gradients of its parameters
=====================================  ==================================================================================

Each has a fake implementation (shapes and dtypes only), so Dynamo + AOT autograd trace THROUGH them: the small task compiles with
``fullgraph=True`` (``aot_eager``) and trains to bit-identical parameters (tests/test_graph_capture_gpu.py).  The loss is an operator
with Python state (which pairs matched, the packed operands, workspaces): the forward op parks its run in a small registry and
returns a token tensor that the backward op redeems; a run whose backward never comes (evaluation) is evicted by the next ones.

An encoder patched by ``mmlearn_amd.fused.accelerate_encoder`` is ONE operator (``mmlearn_amd::tower_fwd`` / ``tower_bwd``, below): its
kernels are not traced one by one -- the launch-free form of those steps is HIP-graph capture (``mmlearn_amd.graph``).
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


# ------------------------------------------------------------------------------------------------ accelerated towers
# An encoder that ``mmlearn_amd.fused.accelerate_encoder`` has patched runs dozens of autograd Functions over ctypes kernels.  Under
# ``torch.compile`` the task hands such a tower to the tracer as ONE operator: the forward op runs the tower's own (eager) forward with
# autograd on and parks the output's graph, the backward op differentiates that graph for the parameters (and for inputs that asked for
# it).  The parameters are operands of the forward op, so their gradients arrive through the compiled graph like any other.  Nothing
# inside the tower is traced or changed -- which is the point: the kernels are the fast path already, and ``fullgraph=True`` holds.
_TOWERS: "weakref.WeakValueDictionary[int, Any]" = weakref.WeakValueDictionary()
_tower_counter = itertools.count(1)
_TOWER_RUNS: "OrderedDict[int, Any]" = OrderedDict()


def mark_tower(module) -> None:
    """Called by ``accelerate_encoder``: this module's forward reaches HIP kernels the tracer cannot see into.  Call it again on a
    (deep) COPY of such a module: the copy carries its original's key."""
    tid = module.__dict__.get("_mmk_tower_id")
    if tid is None or _TOWERS.get(tid) is not module:
        tid = next(_tower_counter)
        module.__dict__["_mmk_tower_id"] = tid
        module.__dict__.pop("_mmk_out_meta", None)
        _TOWERS[tid] = module


def is_opaque_tower(module) -> bool:
    """(plain attribute reads only: the tracer evaluates this while tracing)"""
    return getattr(module, "_mmk_tower_id", None) is not None


def _flatten_inputs(inputs: dict, prefix: str = ""):
    """(key paths of the tensor leaves, the tensors, the other leaves as a JSON-able dict) of a (nested) batch dict."""
    keys, tensors, consts = [], [], {}
    for k, v in inputs.items():
        path = f"{prefix}{k}"
        if isinstance(v, Tensor):
            keys.append(path)
            tensors.append(v)
        elif isinstance(v, dict):
            kk, tt, cc = _flatten_inputs(v, path + "/")
            keys += kk
            tensors += tt
            consts.update(cc)
        elif v is None or isinstance(v, (bool, int, float, str)):
            consts[path] = v
        # anything else (python objects a tower does not read) stays outside the operator
    return keys, tensors, consts


def _rebuild_inputs(keys: Sequence[str], tensors: Sequence[Tensor], consts: dict) -> dict:
    out: dict = {}
    for path, v in itertools.chain(zip(keys, tensors), consts.items()):
        d = out
        parts = path.split("/")
        for part in parts[:-1]:
            d = d.setdefault(part, {})
        d[parts[-1]] = v
    return out


def _encode_tower_meta(keys: Sequence[str], consts: dict, grad_inputs: Sequence[int]) -> str:
    """The non-tensor part of a tower call as one string operand (built with plain string operations: the tracer runs this)."""
    # separators are control characters (unit / record / group separator): keys and string values are free to hold '|', '=', ':'
    cs = []
    for path, v in consts.items():
        if v is None:
            cs.append(path + "\x1dn\x1d")
        elif isinstance(v, bool):
            cs.append(path + "\x1db\x1d" + ("1" if v else "0"))
        elif isinstance(v, int):
            cs.append(path + "\x1di\x1d" + str(v))
        elif isinstance(v, float):
            cs.append(path + "\x1df\x1d" + repr(v))
        else:
            cs.append(path + "\x1ds\x1d" + v)
    return "\x1e".join(keys) + "\x1f" + "\x1e".join(cs) + "\x1f" + ",".join(str(i) for i in grad_inputs)


def _decode_tower_meta(meta: str):
    ks, cs, gi = meta.split("\x1f")
    consts = {}
    for item in filter(None, cs.split("\x1e")):
        path, t, v = item.split("\x1d", 2)
        consts[path] = None if t == "n" else (v == "1") if t == "b" else int(v) if t == "i" else float(v) if t == "f" else v
    return {"keys": [k for k in ks.split("\x1e") if k], "consts": consts, "grad_inputs": [int(i) for i in gi.split(",") if i]}


class _autograd_inside_an_operator:
    """An operator's implementation runs BELOW the autograd dispatch keys (they are excluded for the thread): nothing it calls is
    recorded, whatever the grad mode.  The tower operator's forward needs its encoder's own autograd graph -- this puts the keys back
    for the duration (and restores the exclusions after)."""

    _KEYS = tuple(getattr(torch._C.DispatchKey, n) for n in ("AutogradCPU", "AutogradCUDA", "AutogradOther", "ADInplaceOrView", "AutogradFunctionality")
                  if hasattr(torch._C.DispatchKey, n))

    def __enter__(self):
        self._prev = [torch._C._dispatch_tls_is_dispatch_key_excluded(k) for k in self._KEYS]
        for k in self._KEYS:
            torch._C._dispatch_tls_set_dispatch_key_excluded(k, False)

    def __exit__(self, *exc):
        for k, was in zip(self._KEYS, self._prev):
            torch._C._dispatch_tls_set_dispatch_key_excluded(k, was)


def _tower_call(module, keys, tensors, consts, bf16_autocast: bool, cache: bool = True):
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bool(bf16_autocast), cache_enabled=cache):
        return module(_rebuild_inputs(keys, tensors, consts))[0]


@torch.library.custom_op("mmlearn_amd::tower_fwd", mutates_args=())
def _tower_fwd(tensors: List[Tensor], params: List[Tensor], tower: int, meta: str, needs_grad: bool, bf16_autocast: bool) -> Tuple[Tensor, Tensor]:
    module = _TOWERS.get(tower)
    if module is None:
        raise RuntimeError("mmlearn_amd::tower_fwd: the encoder of this compiled graph no longer exists")
    first = next(iter(module.parameters()), None)
    if first is not None and (not params or params[0].data_ptr() != first.data_ptr()):
        raise RuntimeError("mmlearn_amd::tower_fwd: the registered encoder is not the one whose parameters were passed -- a copy of an "
                           "accelerated encoder must be registered again: mmlearn_amd.compiled.mark_tower(copy)")
    m = _decode_tower_meta(meta)
    leaves, ins = [], []
    for i, t in enumerate(tensors):
        if needs_grad and i in m["grad_inputs"]:
            t = t.detach().requires_grad_(True)
            leaves.append(t)
        ins.append(t)
    if needs_grad:
        with _autograd_inside_an_operator(), torch.enable_grad():
            for i, t in enumerate(ins):
                if i in m["grad_inputs"]:
                    t.requires_grad_(True)   # (the flag set above, below the autograd keys, does not make a leaf)
            out = _tower_call(module, m["keys"], ins, m["consts"], bf16_autocast)
    else:
        with torch.no_grad():
            out = _tower_call(module, m["keys"], ins, m["consts"], bf16_autocast)
    tok = 0
    if needs_grad and out.requires_grad:
        tok = next(_run_counter)
        _TOWER_RUNS[tok] = (out, leaves, [p for p in module.parameters() if p.requires_grad])
        while len(_TOWER_RUNS) > MAX_PARKED_RUNS:
            _TOWER_RUNS.popitem(last=False)
    return out.detach(), torch.tensor([tok], dtype=torch.int64)


@_tower_fwd.register_fake
def _(tensors, params, tower, meta, needs_grad, bf16_autocast):
    from torch._subclasses.fake_tensor import unset_fake_temporarily

    module = _TOWERS.get(tower)
    if module is None:
        raise RuntimeError("mmlearn_amd::tower_fwd: unknown encoder")
    sig = (tuple((tuple(t.shape), t.dtype) for t in tensors), bool(bf16_autocast))
    cache = module.__dict__.setdefault("_mmk_out_meta", {})
    if sig not in cache:
        # the output's shape and dtype: learnt once per input signature from ONE real forward on zeros (evaluation mode, no
        # autograd, so no random numbers are drawn and nothing is kept), outside the tracer's fake-tensor mode
        m = _decode_tower_meta(meta)
        with unset_fake_temporarily(), torch.no_grad():
            real = [torch.zeros(tuple(t.shape), dtype=t.dtype, device=t.device) for t in tensors]
            modes = [(mod, mod.training) for mod in module.modules()]   # (restored one by one: a tower may keep parts of itself in eval mode)
            module.eval()
            try:
                # (no autocast weight cache: a bf16 copy cached here, made without autograd, would be what the REAL forward of the
                # same enclosing autocast region picks up -- and the weight's gradient would be lost)
                out = _tower_call(module, m["keys"], real, m["consts"], bf16_autocast, cache=False)
            finally:
                for mod, was in modes:
                    mod.training = was
            cache[sig] = (tuple(out.shape), out.dtype, out.device)
    shape, dt, dev = cache[sig]
    return torch.empty(shape, dtype=dt, device=dev), torch.empty((1,), dtype=torch.int64, device="cpu")


@torch.library.custom_op("mmlearn_amd::tower_bwd", mutates_args=())
def _tower_bwd(token: Tensor, grad_out: Tensor, grad_inputs: List[Tensor], grad_params: List[Tensor]) -> Tuple[List[Tensor], List[Tensor]]:
    """grad_inputs / grad_params: the operands that receive a gradient (only their shapes are used: the outputs look like them)."""
    run = _TOWER_RUNS.pop(int(token[0]), None)
    if run is None:
        raise RuntimeError("mmlearn_amd::tower_bwd: no parked forward run for this token (backward called twice, or more than "
                           f"{MAX_PARKED_RUNS} forward passes without a backward in between)")
    out, leaves, ps = run
    with _autograd_inside_an_operator():
        grads = torch.autograd.grad([out], leaves + ps, [grad_out.to(out.dtype)], allow_unused=True)
    # an operator's outputs may alias neither its inputs nor each other: gradients that are views (the q / k / v slices of one packed
    # weight gradient, an expanded bias gradient) or that share a buffer with the incoming gradient are copied out
    seen = {grad_out.untyped_storage().data_ptr()}

    def own(g, like):
        if g is None:
            return torch.zeros_like(like)
        key = g.untyped_storage().data_ptr()
        if g._base is not None or key in seen or not g.is_contiguous():
            g = g.clone(memory_format=torch.contiguous_format)
            key = g.untyped_storage().data_ptr()
        seen.add(key)
        return g

    gl = [own(g, t) for g, t in zip(grads[:len(leaves)], leaves)]
    gp = [own(g, p) for g, p in zip(grads[len(leaves):], ps)]
    return gl, gp


@_tower_bwd.register_fake
def _(token, grad_out, grad_inputs, grad_params):
    return [torch.empty_like(t) for t in grad_inputs], [torch.empty_like(p) for p in grad_params]


def _tower_setup(ctx, inputs, output):
    tensors, params, _tower, meta, _needs_grad, _bf16 = inputs
    gi = _decode_tower_meta(meta)["grad_inputs"]
    ctx.grad_inputs = gi
    ctx.grad_params = [i for i, p in enumerate(params) if p.requires_grad]
    ctx.n_tensors, ctx.n_params = len(tensors), len(params)
    ctx.save_for_backward(output[1], *[tensors[i] for i in gi], *[params[i] for i in ctx.grad_params])


def _tower_backward(ctx, gout, _gtoken):
    token, *rest = ctx.saved_tensors
    gi_t, gp_t = rest[:len(ctx.grad_inputs)], rest[len(ctx.grad_inputs):]
    gl, gp = _tower_bwd(token, gout, list(gi_t), list(gp_t))
    g_tensors = [None] * ctx.n_tensors
    for i, g in zip(ctx.grad_inputs, gl):
        g_tensors[i] = g
    g_params = [None] * ctx.n_params
    for i, g in zip(ctx.grad_params, gp):
        g_params[i] = g
    return g_tensors, g_params, None, None, None, None


_tower_fwd.register_autograd(_tower_backward, setup_context=_tower_setup)


def tower_forward(module, inputs: dict) -> Tensor:
    """The traced form of ``module(inputs)[0]`` for an accelerated encoder: one operator in the graph."""
    keys, tensors, consts = _flatten_inputs(inputs)
    params = list(module.parameters())
    grad_on = torch.is_grad_enabled()
    grad_inputs = [i for i, t in enumerate(tensors) if grad_on and t.is_floating_point() and t.requires_grad]
    needs_grad = grad_on and (bool(grad_inputs) or any(p.requires_grad for p in params))
    meta = _encode_tower_meta(keys, consts, grad_inputs)
    bf16_autocast = torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
    out, _token = _tower_fwd(tensors, params, module._mmk_tower_id, meta, needs_grad, bf16_autocast)
    return out
