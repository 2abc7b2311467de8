"""Whole training steps as HIP graphs (``torch.cuda.CUDAGraph`` on ROCm = hipGraph): the launch-bound regime of small batches and
deep, narrow towers, where a step is hundreds of launches of a few microseconds each.  The reference plumbs ``torch.compile`` at
the same point (mmlearn/cli/run.py:139); this package's kernels are reached through ctypes, which a tracing compiler does not see,
so the launch-free step here is a captured one.

:class:`CapturedIJEPAStep` -- one I-JEPA pretraining step (teacher forward, context encoder, predictor, fused target + loss,
backward, optimizer, EMA) per call.  What a graph cannot hold is moved out of it:

* the masks: host RNG + upload happen AHEAD of the step (``IJEPA.stage_masks`` / ``commit_masks``: pinned staging, side stream,
  static per-geometry device buffers).  The block geometry varies from step to step (mmlearn/datasets/processors/masking.py:340-352:
  30 .. 42 predicted patches), so there is one graph per geometry ``(context patches, predicted patches)``, captured on first use;
  all graphs share one memory pool (only one replays at a time);
* the optimizer's step count and learning rate: device words (``mmlearn_amd.optim.AdamW(capturable=True)`` or
  ``torch.optim.AdamW(capturable=True)``);
* the EMA decay schedule: a device word, advanced on the host after every replay (``ExponentialMovingAverage.advance``).

Every call is exactly one training step -- the first ``warmup`` calls run it eagerly (they build the optimizer / EMA tables and the
static gradient buffers a capture needs), later ones replay.  Not capturable: dropout inside the towers (its seeds are drawn on the
host per call and would be frozen into the graph) -- refused at construction.
"""

from __future__ import annotations

from typing import Any, Optional

import torch


class CapturedIJEPAStep:
    def __init__(self, task, optimizer: torch.optim.Optimizer, autocast_dtype: Optional[torch.dtype] = torch.bfloat16, warmup: int = 3):
        for m in list(task.encoder.modules()) + list(task.predictor.modules()):
            if isinstance(m, torch.nn.Dropout) and m.p > 0 and m.training:
                raise ValueError("CapturedIJEPAStep: a captured step would replay one dropout pattern; build the towers with dropout 0")
        for g in optimizer.param_groups:
            if not g.get("capturable", False):
                raise ValueError("CapturedIJEPAStep needs an optimizer built with capturable=True (step count / learning rate in device words)")
        self.task, self.opt, self.autocast_dtype = task, optimizer, autocast_dtype
        self.warmup_left = max(int(warmup), 1)
        self.graphs: dict = {}
        self.pool = None
        self.static_images: Optional[torch.Tensor] = None
        self.static_batch: Optional[dict] = None
        self.losses: dict = {}
        self.side = None
        task.target_encoder.capturable = True
        self.replays = 0

    # ------------------------------------------------------------------ one step, as plain launches
    def _step(self, batch: dict) -> torch.Tensor:
        self.opt.zero_grad(set_to_none=False)
        if self.autocast_dtype is not None:
            with torch.autocast("cuda", dtype=self.autocast_dtype):
                loss = self.task.training_step(batch, 0)
        else:
            loss = self.task.training_step(batch, 0)
        loss.backward()
        self.opt.step()
        self.task.on_before_zero_grad(self.opt)   # EMA update of the target encoder (tasks/ijepa.py:108-115)
        return loss

    def __call__(self, batch: dict[str, Any]) -> torch.Tensor:
        task = self.task
        name = task.modality.name
        images = batch[name]
        if self.static_images is None or self.static_images.shape != images.shape or self.static_images.dtype != images.dtype:
            if self.graphs:
                raise ValueError(f"CapturedIJEPAStep was captured for images {tuple(self.static_images.shape)}, got {tuple(images.shape)}")
            self.static_images = torch.empty_like(images)
        self.static_images.copy_(images, non_blocking=True)
        self.static_batch = {k: v for k, v in batch.items() if k != name and k != task.modality.mask}
        self.static_batch[name] = self.static_images
        b = images.shape[0]

        if self.warmup_left > 0:        # eager steps: real training steps that also warm every table and buffer a capture needs
            self.warmup_left -= 1
            if self.side is None:
                self.side = torch.cuda.Stream(device=images.device)
            self.side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):   # torch asks for warm-up on a side stream: the capture pool starts clean
                loss = self._step(self.static_batch)
            torch.cuda.current_stream().wait_stream(self.side)
            task.stage_masks(b, images.device)   # the next call's masks, ahead of it
            return loss

        st = task._mask_stage
        if st is None or (st.pending is None and st.ready is None):
            task.stage_masks(b, images.device)
        key = task.commit_masks()
        graph = self.graphs.get(key)
        if graph is None:
            graph = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, pool=self.pool):
                self.losses[key] = self._step(self.static_batch)
            if self.pool is None:
                self.pool = graph.pool()
            self.graphs[key] = graph
        graph.replay()
        self.replays += 1
        # host side of what the replay did on the device
        st = task._mask_stage
        st.ready = None                     # this step's masks are spent
        task.target_encoder.advance()       # num_updates, next annealed decay, the word the next replay reads
        loss = self.losses[key]
        task.log("train/ema_decay", task.target_encoder.decay, prog_bar=True)
        task.log("train/loss", loss, prog_bar=True, sync_dist=True)
        task.stage_masks(b, images.device)  # next step's masks: host RNG + pinned upload while this replay runs
        return loss
