"""``AdamW`` with the update as ONE multi-tensor HIP launch per parameter group (``csrc/ijepa.hip::adamw_kernel``).

Same algorithm and hyper-parameters as ``torch.optim.AdamW`` (decoupled weight decay, bias correction, ``amsgrad`` /
``maximize`` off) for f32 parameters on the GPU; the foreach implementation makes ~7 passes of 15-25 launches over the
parameters (3.7 ms for the 172 M parameters of ViT-B/16 + BERT-base), this one reads and writes every tensor once.
Use it wherever an mmlearn task takes ``optimizer=partial(torch.optim.AdamW, ...)``:
``optimizer=partial(mmlearn_amd.optim.AdamW, lr=..., weight_decay=...)``.  ``state_dict`` keeps torch's layout
(``step``, ``exp_avg``, ``exp_avg_sq`` per parameter).

``capturable=True`` (as in ``torch.optim.AdamW``): the step count and the learning rate live in device words per group, the kernel
forms the bias corrections itself, and ``step()`` neither reads anything back nor bakes a host scalar into the launch -- the whole
training step can then be captured into a HIP graph (``torch.cuda.graph``) and replayed (tests/test_graph_capture_gpu.py).
Per-parameter ``step`` entries of the state are then device tensors, as in torch; a scheduler that changes ``group["lr"]`` is
honoured by eager steps and, between replays of a graph, through ``opt.lr_device(g).fill_(...)`` (a replay reads the word).

Capture needs STATIC gradients: a captured launch bakes in the addresses of the optimizer's device tables (tensor / chunk tables,
the gradient-pointer table, the step and learning-rate words).  Run one eager step first with the gradient buffers the capture
will use (``zero_grad(set_to_none=False)``); ``step()`` under capture refuses to (re)build a table (a rebuild would allocate pinned
memory inside the capture and point the graph at buffers a later eager step could replace), and every table a capture has used is
kept alive for the optimizer's lifetime, so an eager step with other gradient addresses (a ragged last batch,
``zero_grad(set_to_none=True)``) between replays cannot free what the graph still reads.
"""

from __future__ import annotations

import ctypes as C
from typing import Any, Dict, List

import torch

from . import _lib
from .kernels import check, dtype_tag, require_gpu, stream


class _Tensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("numel", C.c_int64)]


class _Chunk(C.Structure):
    _fields_ = [("tensor", C.c_int32), ("pad", C.c_int32), ("offset", C.c_int64)]


def _to_device(raw: bytes, device) -> torch.Tensor:
    return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 capturable: bool = False):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, capturable=bool(capturable)))
        self._tables: Dict[int, Any] = {}
        self._grad_tables: Dict[int, Any] = {}
        self._dev_words: Dict[int, Any] = {}
        self._captured: List[Any] = []   # tables / words whose addresses live inside a captured graph: never freed

    def _keep_for_graph(self, *objs) -> None:
        for o in objs:
            if not any(o is k for k in self._captured):
                self._captured.append(o)

    def load_state_dict(self, state_dict) -> None:
        super().load_state_dict(state_dict)
        self._tables = {}   # the restored moment tensors are new storages
        self._dev_words = {}

    def _table(self, gi: int, plist: List[torch.Tensor]):
        """Static device tables of one group, rebuilt when the set of parameters with gradients changes or when any of the
        storages the table points at has been replaced (``load_state_dict`` swaps the moment tensors: a table keyed on the
        parameters alone would keep updating the freed buffers)."""
        key = tuple((p.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr()) for p in plist)
        tab = self._tables.get(gi)
        if tab is not None and tab["key"] == key:
            return tab
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("mmlearn_amd.optim.AdamW: the parameter / moment tables must exist before capture -- run one eager step "
                               "with the same set of parameters receiving gradients, then capture")
        chunk = _lib.lib().mmk_adamw_chunk_elems()
        tens = (_Tensor * len(plist))()
        chunks = []
        for k, p in enumerate(plist):
            st = self.state[p]
            tens[k].param, tens[k].exp_avg, tens[k].exp_avg_sq, tens[k].numel = p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()
            chunks.extend((k, off) for off in range(0, p.numel(), chunk))
        carr = (_Chunk * len(chunks))()
        for i, (k, off) in enumerate(chunks):
            carr[i].tensor, carr[i].offset = k, off
        dev = plist[0].device
        tab = {"key": key, "tensors": _to_device(bytes(tens), dev), "chunks": _to_device(bytes(carr), dev), "n_chunks": len(chunks)}
        self._tables[gi] = tab
        return tab

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group["params"] if p.grad is not None]
            if not plist:
                continue
            for p in plist:
                require_gpu(p)
                if p.dtype != torch.float32 or not p.is_contiguous() or p.grad.is_sparse:
                    raise RuntimeError("mmlearn_amd.optim.AdamW supports dense contiguous f32 parameters")
                st = self.state[p]
                if st and (st["exp_avg"].device != p.device or st["exp_avg"].dtype != torch.float32 or not st["exp_avg"].is_contiguous()
                           or st["exp_avg_sq"].device != p.device or st["exp_avg_sq"].dtype != torch.float32 or not st["exp_avg_sq"].is_contiguous()):
                    # state restored from a checkpoint in another layout: bring it to what the kernel reads
                    st["exp_avg"] = st["exp_avg"].to(device=p.device, dtype=torch.float32).contiguous()
                    st["exp_avg_sq"] = st["exp_avg_sq"].to(device=p.device, dtype=torch.float32).contiguous()
                if st and not isinstance(st["step"], torch.Tensor):
                    st["step"] = torch.tensor(float(st["step"]), dtype=torch.float32)
                if st and st["step"].is_cuda and not group.get("capturable", False):
                    st["step"] = st["step"].cpu()   # load_state_dict moves the state to the parameter's device; the step counter lives on the host
                if not st:
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device if group.get("capturable", False) else None)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            tab = self._table(gi, plist)
            if group.get("capturable", False):
                self._step_capturable(gi, group, plist, tab)
                continue
            steps = {int(self.state[p]["step"].item()) for p in plist}
            if len(steps) != 1:
                raise RuntimeError("mmlearn_amd.optim.AdamW needs all parameters of a group at the same step")
            step = steps.pop() + 1
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in plist]
            dev = plist[0].device
            gptr = torch.tensor([g.data_ptr() for g in grads], dtype=torch.int64).to(dev, non_blocking=True)
            gdt = torch.tensor([dtype_tag(g.dtype) for g in grads], dtype=torch.int32).to(dev, non_blocking=True)
            b1, b2 = group["betas"]
            check(_lib.lib().mmk_adamw_update(tab["tensors"].data_ptr(), gptr.data_ptr(), gdt.data_ptr(), tab["chunks"].data_ptr(), tab["n_chunks"],
                                              float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), step,
                                              stream()))
            for p in plist:
                self.state[p]["step"] += 1
        return loss

    def _step_capturable(self, gi: int, group, plist: List[torch.Tensor], tab) -> None:
        """One launch with the step count and the learning rate in device words; nothing here depends on a host read, and every
        host-to-device transfer it needs is made from pinned memory that outlives the launch (the gradient table is rebuilt only
        when the gradients' addresses change).  All parameters of the group share ONE step word: their ``state[p]["step"]`` are
        0-dim views of it, so the per-parameter layout of torch's state_dict costs one increment, not one launch per tensor."""
        dev = plist[0].device
        capturing = torch.cuda.is_current_stream_capturing()
        words = self._dev_words.get(gi)
        if words is None or any(self.state[p]["step"].data_ptr() != words["step"].data_ptr() for p in plist):
            if capturing:
                raise RuntimeError("mmlearn_amd.optim.AdamW(capturable=True): run one eager step before capturing")
            steps = {float(self.state[p]["step"]) for p in plist}      # (host read: first step / after load_state_dict only)
            if len(steps) != 1:
                raise RuntimeError("mmlearn_amd.optim.AdamW needs all parameters of a group at the same step")
            words = {"step": torch.full((1,), steps.pop(), dtype=torch.float32, device=dev),
                     "lr": torch.full((1,), float(group["lr"]), dtype=torch.float32, device=dev), "lr_seen": float(group["lr"])}
            self._dev_words[gi] = words
            for p in plist:
                self.state[p]["step"] = words["step"].view(())
        if not capturing and float(group["lr"]) != words["lr_seen"]:   # a scheduler moved the float: follow it (eager steps only)
            words["lr"].fill_(float(group["lr"]))
            words["lr_seen"] = float(group["lr"])
        if capturing and not all(p.grad.is_contiguous() for p in plist):
            raise RuntimeError("mmlearn_amd.optim.AdamW(capturable=True): non-contiguous gradients cannot be captured")
        grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in plist]
        key = tuple((g.data_ptr(), g.dtype) for g in grads)
        gt = self._grad_tables.get(gi)
        if gt is None or gt["key"] != key:
            if capturing:
                raise RuntimeError("mmlearn_amd.optim.AdamW(capturable=True): the gradients' addresses differ from the last eager step's. "
                                   "A captured step needs static gradient buffers: call zero_grad(set_to_none=False) and run one eager "
                                   "step with those buffers before capturing")
            # pinned staging that lives as long as the table: a captured copy node re-reads it at every replay
            pin_p = torch.tensor([g.data_ptr() for g in grads], dtype=torch.int64).pin_memory()
            pin_t = torch.tensor([dtype_tag(g.dtype) for g in grads], dtype=torch.int32).pin_memory()
            gt = {"key": key, "pin": (pin_p, pin_t), "gptr": pin_p.to(dev, non_blocking=True), "gdt": pin_t.to(dev, non_blocking=True)}
            self._grad_tables[gi] = gt
        if capturing:
            self._keep_for_graph(tab, gt, words)
        words["step"].add_(1.0)
        b1, b2 = group["betas"]
        check(_lib.lib().mmk_adamw_update_dev(tab["tensors"].data_ptr(), gt["gptr"].data_ptr(), gt["gdt"].data_ptr(), tab["chunks"].data_ptr(),
                                              tab["n_chunks"], words["lr"].data_ptr(), float(b1), float(b2), float(group["eps"]),
                                              float(group["weight_decay"]), words["step"].data_ptr(), stream()))

    def lr_device(self, group_index: int = 0) -> torch.Tensor:
        """The device word a captured step reads its learning rate from (``capturable=True``, after the first eager step):
        ``opt.lr_device(g).fill_(new_lr)`` between replays is what a scheduler's ``group["lr"] = ...`` is for eager steps."""
        return self._dev_words[group_index]["lr"]
