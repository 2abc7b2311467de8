"""Headline benchmark: image-text pairs/s of one contrastive-pretraining step.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1] per GPU; configs[2] at N = 8): CLIP ViT-B/16 image encoder
(HF ``CLIPVisionModelWithProjection`` from config, projection 512) + BERT-base text encoder (HF
``BertModel`` from config + Linear(768, 512)), random-init weights, synthetic batches
(``rand(B,3,224,224)`` pixels, ``randint(0, 30522, (B,77))`` tokens, fully paired ids), bf16 autocast,
per-GPU batch 1024, AdamW(lr 1e-4, weight decay 0.1).  One step = encoders forward -> HIP L2-normalise -> HIP
contrastive loss (global-batch negatives via the packed all-gather when N > 1) -> backward -> optimizer step.  The
encoders are the stock HF modules with the kernels of SURVEY 8(f1) swapped in underneath (``mmlearn_amd.fused``:
same parameters, same math; ``--no-fused-encoder-ops`` runs them untouched, 421 ms vs ~215 ms per step).

The JSON line carries, besides the driver contract:
  roofline     -- the dominant MFMA kernel of the loss path, timed with HIP events on its launch stream
                  inside the timed region, against the dense bf16 MFMA peak (2.5 PFLOP/s);
  roofline_widened -- the same for the dominant hand-written kernel of the whole step (the weight-gradient GEMM);
  cpu_baseline -- the same step (same model, torch eager ops, oracle/eager_torch loss = the reference's op
                  sequence) on rank 0's host cores for a bounded sample (at every N);
  roofline_shard -- one rank's share of the row-sharded loss at configs[2] (R = 1024 x C = 8192), timed at N = 1.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from functools import partial

import torch
import torch.distributed as dist
from torch import nn

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBPS = 8000.0          # HBM3E, same guide


class VisionEncoder(nn.Module):
    """HF CLIP ViT-B/16 with projection; mmlearn encoder contract: forward(dict) -> (embedding,)"""

    mmk_reads_only_token0 = True   # forward reads .image_embeds = projection(post_layernorm(last_hidden_state[:, 0])) and nothing else

    def __init__(self, small: bool = False, hip_attention: bool = False):
        super().__init__()
        from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection

        cfg = CLIPVisionConfig(patch_size=16, image_size=224, projection_dim=512, hidden_size=768, intermediate_size=3072,
                               num_hidden_layers=12, num_attention_heads=12)
        if small:
            cfg = CLIPVisionConfig(patch_size=32, image_size=224, projection_dim=512, hidden_size=128, intermediate_size=256,
                                   num_hidden_layers=2, num_attention_heads=2)
        if hip_attention:  # whole-sequence-on-chip attention kernel through HF's AttentionInterface
            from mmlearn_amd.attention import register_hf_attention

            cfg._attn_implementation = register_hf_attention()
        self.model = CLIPVisionModelWithProjection(cfg)

    def forward(self, inputs):
        return (self.model(pixel_values=inputs["rgb"]).image_embeds,)


class TextEncoder(nn.Module):
    """HF BERT-base (no pooler) + CLS token + Linear(768, 512)."""

    mmk_reads_only_token0 = True   # forward reads last_hidden_state[:, 0] and nothing else

    def __init__(self, small: bool = False, hip_attention: bool = False):
        super().__init__()
        from transformers import BertConfig, BertModel

        cfg = BertConfig()
        if small:
            cfg = BertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256)
        if hip_attention:  # same hook as the ViT; BERT's attention-probability dropout (0.1) runs inside the kernel
            from mmlearn_amd.attention import register_hf_attention

            cfg._attn_implementation = register_hf_attention()
        self.model = BertModel(cfg, add_pooling_layer=False)
        self.proj = nn.Linear(cfg.hidden_size, 512, bias=False)

    def forward(self, inputs):
        # the tokenizer's mask rides along as the reference's text towers forward it (mmlearn/modules/encoders/text.py:160-165)
        h = self.model(input_ids=inputs["text"], attention_mask=inputs.get("attention_mask")).last_hidden_state[:, 0]
        return (self.proj(h),)


class _Step(nn.Module):
    """DDP wraps forward(); route it to training_step like Lightning's strategy wrapper does."""

    def __init__(self, task):
        super().__init__()
        self.task = task

    def forward(self, batch):
        return self.task.training_step(batch, 0)


def synthetic_batch(b: int, rank: int, device, padded: bool = False):
    """random pixels, random tokens "with an all-ones attention mask", fully paired ids.  ``padded``: caption lengths
    ~ U[8, 77] instead (right padding, the tokenizer's form) -- the `padded_text` leg."""
    g = torch.Generator(device="cpu").manual_seed(1000 + rank)
    ids = torch.stack([torch.zeros(b, dtype=torch.long), torch.arange(rank * b, (rank + 1) * b)], 1)
    batch = {
        "rgb": torch.rand(b, 3, 224, 224, generator=g).to(device),
        "text": torch.randint(0, 30522, (b, 77), generator=g).to(device),
        "example_ids": {"rgb": ids.to(device), "text": ids.to(device)},
    }
    mask = torch.ones(b, 77, dtype=torch.long)
    if padded:
        mask = (torch.arange(77)[None, :] < torch.randint(8, 78, (b, 1), generator=g)).long()
    batch["attention_mask"] = mask.to(device)
    if os.environ.get("MMK_BENCH_PAIRED_HINT"):  # A/B: what mmlearn_amd.wire.DefaultDataCollator adds (default: the matcher runs)
        batch["fully_paired"] = True
    return batch


def _adamw(fused: bool):
    """AdamW(lr=1e-4, weight_decay=0.1): torch's, or the same update as one HIP launch per group (mmlearn_amd.optim)."""
    if fused and os.environ.get("MMK_BENCH_TORCH_ADAMW") is None:
        from mmlearn_amd.optim import AdamW

        return partial(AdamW, lr=1e-4, weight_decay=0.1)
    return partial(torch.optim.AdamW, lr=1e-4, weight_decay=0.1)


def build_task(loss, small: bool, fused: bool = False, cls_only="auto"):
    from mmlearn_amd.tasks.contrastive_pretraining import ContrastivePretraining

    torch.manual_seed(0)
    rgb, text = VisionEncoder(small, hip_attention=fused), TextEncoder(small, hip_attention=fused)
    if fused:  # HIP LayerNorm / quick-GELU inside the encoders (same parameters, same math)
        from mmlearn_amd.fused import accelerate_encoder

        # CLIP is pre-LN: these LayerNorms feed autocast Linears only, so they may emit bf16 directly;
        # BERT is post-LN (the LN output is the residual stream) and keeps f32 outputs.
        add_ln = os.environ.get("MMK_BENCH_NO_ADD_LN") is None   # A/B switch for the fused residual add + LayerNorm
        accelerate_encoder(rgb, low_precision_ln=("layer_norm1", "layer_norm2", "post_layernorm"), fuse_qkv=True, fuse_add_ln=add_ln,
                           cls_only=cls_only)
        accelerate_encoder(text, fuse_qkv=True, fuse_add_ln=add_ln, cls_only=cls_only)
    task = ContrastivePretraining(
        encoders={"rgb": rgb, "text": text},
        loss=loss,
        optimizer=_adamw(fused),
        compute_validation_loss=False,
        compute_test_loss=False,
    )
    if fused and os.environ.get("MMK_BENCH_NO_STREAMS") is None:   # one HIP stream per tower (A/B switch: single stream)
        task.concurrent_encoders = True
    return task


def cpu_baseline(seconds_budget: float = 25.0):
    """The reference's step on the host cores: same encoders (f32), eager torch ops, the reference's loss op
    sequence (oracle/eager_torch.py).  Bounded sample: batches of 32 pairs until the time budget is spent (at most 32 threads;
    round 2 sampled batches of 8, which under-fed the host GEMMs: a batch-8 figure next to a batch-1024 one)."""
    from oracle.eager_torch import EagerContrastiveLoss

    import mmlearn_amd.tasks.contrastive_pretraining as cp

    # many-core hosts: torch eager on ~200 threads is slower than on 32 (measured on the 256-core MI355X box:
    # 16 pairs took ~7 min at 256 threads); use at most 32 and report what was used
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    b = 32
    saved = cp.l2_normalize
    cp.l2_normalize = lambda x: torch.nn.functional.normalize(x, p=2, dim=-1)  # the reference's K1 on CPU (baseline leg only)
    try:
        task = build_task(EagerContrastiveLoss(), small=False)
        opt = task.configure_optimizers()
        batch = synthetic_batch(b, 0, torch.device("cpu"))

        def step():
            opt.zero_grad(set_to_none=True)
            loss = task.training_step(batch, 0)
            loss.backward()
            opt.step()

        t0 = time.perf_counter()
        step()  # first step (includes one-time allocator / thread-pool start-up)
        first = time.perf_counter() - t0
        n, dt = 1, first
        if first < seconds_budget / 2:  # time steady-state steps while the budget lasts
            t0 = time.perf_counter()
            n = 0
            while n < 1 or ((time.perf_counter() - t0) + first < seconds_budget and n < 50):
                step()
                n += 1
            dt = time.perf_counter() - t0
    finally:
        cp.l2_normalize = saved
    return {"value": round(b * n / dt, 3), "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"{n} steps of {b} pairs (ViT-B/16 + BERT-base, f32, torch eager on {cores} host threads, reference loss op sequence)"}



class _gc_quiet:
    """Microsecond-scale wall clocks only (the loss-path legs): a generation-2 pass of Python's collector over a process that has
    imported torch + transformers takes ~60 ms -- measured in round 5 as ONE 63 ms iteration among twenty 120 us ones, which is where
    round 4's "2,640 us wall per rank share" came from.  Collect first, keep the collector off inside the timed loop (what `timeit`
    does), switch it back on afterwards.  The headline step (190 ms of device work queued ahead) is timed with the collector on."""

    def __enter__(self):
        import gc

        gc.collect()
        self.was = gc.isenabled()
        gc.disable()

    def __exit__(self, *exc):
        import gc

        if self.was:
            gc.enable()


def _timed_steps(step, warmup: int, steps: int) -> float:
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def eager_gpu_leg(batch_size: int, rank: int, dev, small: bool, steps: int = 8, warmup: int = 2, padded: bool = False):
    """The denominator of the north-star ratio, on the driver's clock: the SAME step as the reference runs it on
    PyTorch-ROCm -- stock HF encoders (SDPA attention, hipBLASLt, ATen elementwise), torch.optim.AdamW, bf16 autocast and
    the reference's loss op sequence (oracle/eager_torch.py = contrastive.py:134-144,327-340) -- same batch, this GPU,
    local negatives.  Baseline leg only: nothing of it is in ``value``."""
    from oracle.eager_torch import EagerContrastiveLoss

    import mmlearn_amd.tasks.contrastive_pretraining as cp

    saved = cp.l2_normalize
    cp.l2_normalize = lambda x: torch.nn.functional.normalize(x, p=2, dim=-1)   # the reference's F.normalize
    try:
        task = build_task(EagerContrastiveLoss(), small, fused=False).to(dev)
        opt = task.configure_optimizers()
        batch = synthetic_batch(batch_size, rank, dev, padded=padded)

        def step():
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = task.training_step(batch, 0)
            loss.backward()
            opt.step()
            return loss

        sec = _timed_steps(step, warmup, steps)
        final = float(step().detach().float())
    finally:
        cp.l2_normalize = saved
    peak = torch.cuda.max_memory_allocated() / 2**30
    del task, opt
    torch.cuda.empty_cache()
    return {"ms_per_step": round(sec * 1e3, 2), "pairs_s": round(batch_size / sec, 1), "steps": steps, "warmup": warmup, "per_gpu_batch": batch_size,
            "peak_hbm_gib": round(peak, 1), "loss": round(final, 4),
            "text_mask": "caption lengths ~U[8, 77] (right padding)" if padded else "all ones (SURVEY 8(d))",
            "what": "stock HF CLIP ViT-B/16 + BERT-base, SDPA, hipBLASLt, torch.optim.AdamW, bf16 autocast, reference loss op sequence (eager), 1 GPU, local negatives"}


def padded_text_leg(b: int, dev, small: bool, steps: int = 8, warmup: int = 3):
    """The headline step on a text batch with REAL padding -- caption lengths ~U[8, 77], the mask the tokenizer emits and the
    reference's text towers forward (mmlearn/modules/encoders/text.py:160-165): every BERT layer then runs the masked HIP attention
    kernels (csrc/attention.hip MASK variants, the single-query kernel in the last layer), never the library's SDPA; the launches are
    counted to show it.  NOT part of `value`; the stock step on the same batch is the `padded_text_eager` leg."""
    from mmlearn_amd import ContrastiveLoss
    from mmlearn_amd import kernels as K

    task = build_task(ContrastiveLoss(static_shapes=True), small, fused=True).to(dev)
    opt = task.configure_optimizers()
    batch = synthetic_batch(b, 0, dev, padded=True)
    seen = {"attn_fwd_masked": 0, "attn_fwd_unmasked": 0, "cls_attn_masked": 0, "cls_attn_unmasked": 0, "stock_attention_forwards": 0}
    real_fwd, real_cls = K.attn_fwd, K.cls_attn_fwd

    def fwd(q, k, v, scale, dropout_p=0.0, seed=0, key_bias=None, causal=False):
        seen["attn_fwd_masked" if key_bias is not None else "attn_fwd_unmasked"] += 1
        return real_fwd(q, k, v, scale, dropout_p, seed, key_bias, causal)

    def cls(q, kv, scale, dropout_p=0.0, seed=0, key_bias=None):
        seen["cls_attn_masked" if key_bias is not None else "cls_attn_unmasked"] += 1
        return real_cls(q, kv, scale, dropout_p, seed, key_bias)

    def counting(orig):
        def stock(*a, **k):
            seen["stock_attention_forwards"] += 1
            return orig(*a, **k)
        return stock

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = task.training_step(batch, 0)
        loss.backward()
        opt.step()
        return loss

    sec = _timed_steps(step, warmup, steps)
    K.attn_fwd, K.cls_attn_fwd = fwd, cls
    for m in task.modules():
        if hasattr(m, "_mmk_stock_forward") and type(m).__name__ in ("BertSelfAttention", "CLIPAttention"):
            m._mmk_stock_forward = counting(m._mmk_stock_forward)
    try:
        loss = step()
        torch.cuda.synchronize()
    finally:
        K.attn_fwd, K.cls_attn_fwd = real_fwd, real_cls
    return {"workload": f"the headline step (configs[1], per-GPU batch {b}) on captions of length ~U[8, 77] with the tokenizer's padding mask",
            "ms_per_step": round(sec * 1e3, 2), "pairs_s": round(b / sec, 1), "steps": steps, "warmup": warmup,
            "loss": round(float(loss.detach().float()), 4),
            "attention_launches_in_one_step": seen,
            "note": "text layers 1-11 = attn_fwd_masked, the token-0 last layer = cls_attn_masked; the image tower has no mask"}


class _PooledVision(nn.Module):
    def __init__(self, small):
        super().__init__()
        from transformers import CLIPVisionConfig, CLIPVisionModel
        from mmlearn_amd.attention import register_hf_attention

        cfg = CLIPVisionConfig(patch_size=16, image_size=224, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12)
        if small:
            cfg = CLIPVisionConfig(patch_size=32, image_size=224, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2)
        cfg._attn_implementation = register_hf_attention()
        self.model = CLIPVisionModel(cfg)

    def forward(self, inputs):
        return (self.model(pixel_values=inputs["rgb"]).pooler_output,)


class _PooledText(nn.Module):
    def __init__(self, small):
        super().__init__()
        from transformers import BertConfig, BertModel
        from mmlearn_amd.attention import register_hf_attention

        cfg = BertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256) if small else BertConfig()
        cfg._attn_implementation = register_hf_attention()
        self.model = BertModel(cfg, add_pooling_layer=False)

    def forward(self, inputs):
        return (self.model(input_ids=inputs["text"], attention_mask=inputs.get("attention_mask")).last_hidden_state[:, 0],)


class _PooledAudio(nn.Module):
    """HTSAT (Swin-style audio transformer) as shipped in HF CLAP, from config; stock HF ops (windowed attention with a
    relative-position bias is outside what the HIP attention kernel serves)."""

    def __init__(self, small):
        super().__init__()
        from transformers import ClapAudioConfig, ClapAudioModel

        cfg = ClapAudioConfig(depths=(1, 1, 1, 1), patch_embeds_hidden_size=16, hidden_size=128) if small else ClapAudioConfig()
        self.model = ClapAudioModel(cfg)
        self.width = cfg.hidden_size

    def forward(self, inputs):
        x = inputs["audio"]
        return (self.model(input_features=x, is_longer=torch.zeros(x.shape[0], 1, dtype=torch.bool, device=x.device)).pooler_output,)


def full_last_layer_leg(b: int, dev, small: bool, steps: int = 8, warmup: int = 3):
    """The headline step with ``accelerate_encoder(..., cls_only=False)`` (NOT part of `value`; round 4's headline).  Both towers are
    pooled at token 0 (mmlearn/modules/encoders/clip.py:463-470; the text tower's ``last_hidden_state[:, 0]``) and say so
    (``mmk_reads_only_token0``), so ``accelerate_encoder``'s default (``cls_only="auto"``) lets the last layer of each compute keys /
    values for all tokens and everything else for token 0 only -- same loss, same gradient for every parameter
    (tests/test_cls_only_cpu.py, tests/test_fused_gpu.py), 10/12 of the last layer's GEMM work never issued.  This leg switches
    that off, so the line shows both."""
    from mmlearn_amd import ContrastiveLoss

    loss_fn = ContrastiveLoss(static_shapes=True)
    task = build_task(loss_fn, small, fused=True, cls_only=False).to(dev)
    opt = task.configure_optimizers()
    batch = synthetic_batch(b, 0, dev)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = task.training_step(batch, 0)
        loss.backward()
        opt.step()
        return loss

    sec = _timed_steps(step, warmup, steps)
    loss = step()
    torch.cuda.synchronize()
    return {"workload": f"the headline step (configs[1], per-GPU batch {b}) with the last layer of both towers computed for ALL tokens "
                        "(accelerate_encoder(cls_only=False); loss and all parameter gradients are those of the headline step)",
            "ms_per_step": round(sec * 1e3, 2), "pairs_s": round(b / sec, 1), "steps": steps, "warmup": warmup,
            "loss": round(float(loss.detach().float()), 4)}


def three_tower_leg(b: int, dev, small: bool, steps: int = 8, warmup: int = 2, stock: bool = False):
    """BASELINE configs[3] on the driver's clock (bounded, outside the headline region): image + text + audio (HTSAT)
    towers, ONE shared projection head (Linear 768 -> 512), learnable logit scale, three weighted loss pairs -> the
    N-way pairwise similarity path (3 pairs = 6 CE directions in one launch set).  The bioscan_1m recipe shape
    (projects/bioscan_clip/configs/experiment/bioscan_1m.yaml:11-15)."""
    from mmlearn_amd import ContrastiveLoss, _lib
    from mmlearn_amd.fused import accelerate_encoder
    from mmlearn_amd.tasks import ContrastivePretraining, LossPairSpec, ModuleKeySpec

    if os.environ.get("MMK_BENCH_TILED_LOSS") is not None:   # A/B switch: the tiled multi-launch loss instead of the one-launch kernel
        from mmlearn_amd import kernels as _K

        _K.FUSED_LOSS = False
    torch.manual_seed(0)
    rgb, text, audio = _PooledVision(small), _PooledText(small), _PooledAudio(small)
    if stock:   # the denominator: the same three towers as PyTorch-ROCm runs them, torch AdamW, the reference's loss op sequence
        return _three_tower_stock(rgb, text, audio, b, dev, small, steps, warmup)
    for m, tower in (("rgb", rgb), ("text", text), ("audio", audio)):
        accelerate_tower(tower, m)
    width = 128 if small else 768
    task = ContrastivePretraining(
        encoders={"rgb": rgb, "text": text, "audio": audio},
        heads={"shared": {"proj": nn.Linear(width, 512, bias=False)}},
        modality_module_mapping={m: ModuleKeySpec(encoder_key=m, head_key="shared") for m in ("rgb", "text", "audio")},
        loss=ContrastiveLoss(), optimizer=_adamw(True),
        modality_loss_pairs=[LossPairSpec(("rgb", "text")), LossPairSpec(("rgb", "audio"), 0.5), LossPairSpec(("text", "audio"), 0.5)],
        compute_validation_loss=False, compute_test_loss=False).to(dev)
    task.concurrent_encoders = True
    opt = task.configure_optimizers()
    batch = _three_tower_batch(b, dev)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = task.training_step(batch, 0)
        loss.backward()
        opt.step()
        return loss

    sec = _timed_steps(step, warmup, steps)
    # per-kernel events of the loss path: one single-stream pass
    task.concurrent_encoders, task.match_ahead = False, False
    step()
    torch.cuda.synchronize()
    _lib.profile_read()
    _lib.profile_enable(True)
    loss = step()
    torch.cuda.synchronize()
    prof = _lib.profile_read()
    _lib.profile_enable(False)
    # where the step goes: forward + backward of each tower alone (same batch, one stream, 3 passes each after one warm-up)
    from mmlearn_amd.modalities import Modalities

    tower_ms = {}
    for m in ("rgb", "text", "audio"):
        def tower_pass(m=m):
            task.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                e = task.encode(batch, Modalities.get_modality(m), normalize=True)
            e.float().sum().backward()
        tower_ms[m] = round(_timed_steps(tower_pass, 1, 3) * 1e3, 2)
    out = {"workload": f"BASELINE configs[3] on 1 GPU: CLIP ViT-B/16 + BERT-base + HTSAT (HF CLAP audio), shared Linear({width},512) head, 3 weighted pairs, per-GPU batch {b}, bf16",
           "tower_fwd_bwd_ms": tower_ms,
           "ms_per_step": round(sec * 1e3, 2), "samples_s": round(b / sec, 1), "steps": steps, "warmup": warmup, "loss": round(float(loss.detach().float()), 4),
           "roofline": _loss_roofline(prof, n_rows=b, n_cols=b, d=512, n_pairs=3, steps=1)}
    wa = _window_attn_roofline(prof, audio, b)
    if wa is not None:
        out["roofline_widened"] = wa
    del task, opt, batch
    torch.cuda.empty_cache()
    return out


def _window_attn_roofline(prof: dict, audio, b: int):
    """HBM roofline of the windowed-attention kernels of the audio tower (csrc/window_attention.hip) over one step: algorithmic bytes --
    per window and head q, k, v in and o out forward, q, k, v, dO in and dq, dk, dv out backward, [64 x head_dim] bf16 each, + lse --
    summed over the tower's layers, over the kernels' dispatch-stamped durations.  `traffic` = HBM-side bytes of ONE first-resolution
    launch pair from the committed PMC passes (profiles/r05_window_attn_pmc_traffic.json), with its algorithmic bytes beside it."""
    f, g = prof.get("win_attn_fwd"), prof.get("win_attn_bwd")
    if not f or not g or not f[0] or not g[0]:
        return None
    units = 0       # sum over layers of (window-heads x 64 x head_dim x 2 bytes) = bytes of one [tokens, C] operand
    lse = 0
    for m in audio.modules():
        if hasattr(m, "relative_position_bias_table") and hasattr(m, "all_head_size"):
            rows = getattr(m, "_mmk_rows", None)    # token rows its last fused call attended over (set by the patched forwards)
            if not rows:
                continue
            units += rows * m.all_head_size * 2
            lse += rows * m.num_attention_heads * 4
    if not units:
        return None
    alg = (4 + 7) * units + 2 * lse
    sec = (f[1] + g[1]) * 1e-3
    out = {"bound": "hbm", "kernel": "win_attn_fwd + win_attn_bwd", "achieved": round(alg / sec * 1e-9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
           "frac": round(alg / sec * 1e-9 / HBM_PEAK_GBPS, 4), "traffic": None, "launches": int(f[0] + g[0]),
           "device_us_per_step": round(sec * 1e6, 1), "algorithmic_bytes_per_step": int(alg)}
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r05_window_attn_pmc_traffic.json")))
        out["traffic"] = int(pmc["fwd"]["hbm_bytes_per_launch"] + pmc["bwd"]["hbm_bytes_per_launch"])
        out["traffic_note"] = (f"one forward + backward launch at the first resolution ({pmc['windows_per_sample']} windows x {pmc['heads']} heads, batch "
                               f"{pmc['batch']}): {pmc['fwd']['algorithmic_bytes'] + pmc['bwd']['algorithmic_bytes']} algorithmic bytes")
    except Exception:
        pass
    return out


def accelerate_tower(tower, modality: str):
    """What the HIP leg of configs[3] swaps into each tower (also used by tools/prof_tower.py)."""
    from mmlearn_amd.fused import accelerate_encoder, patch_conv_as_gemm

    if modality == "rgb":
        return accelerate_encoder(tower, low_precision_ln=("layer_norm1", "layer_norm2"), fuse_qkv=True, fuse_add_ln=True)
    if modality == "text":
        return accelerate_encoder(tower, fuse_qkv=True, fuse_add_ln=True)
    # HTSAT: the LayerNorm swap (f32 in / f32 out), every Linear's weight gradient on csrc/wgrad.hip, its windowed attention as
    # one kernel each way (csrc/window_attention.hip; MMK_BENCH_NO_WINDOW_ATTN=1 leaves it on ATen) ...
    swapped = accelerate_encoder(tower, low_precision_ln=("layernorm_before", "layernorm_after"),   # both feed nothing but Linears
                                 wgrad_linear=os.environ.get("MMK_BENCH_NO_AUDIO_WGRAD") is None,
                                 window_attention=os.environ.get("MMK_BENCH_NO_WINDOW_ATTN") is None)
    # ... and its 4 x 4 / stride 4 patch embedding as im2col + GEMM (the image comes out of a BatchNorm: patchify has a backward).
    # Besides the time, this takes MIOpen's implicit-GEMM convolution kernels out of the leg: with AMD_SERIALIZE_KERNEL=3 +
    # AMD_LOG_LEVEL=3 the leg's `Memory access fault by GPU` (round 2: intermittent, unexplained; round 3: reproducible once the
    # allocation pattern changed) is raised by `igemm_bwd_gtcx35_nhwc_bf16_...`, the library's backward kernel of exactly this
    # convolution.
    swapped["patch_conv"] = patch_conv_as_gemm(tower)
    return swapped


def _three_tower_batch(b: int, dev):
    g = torch.Generator(device="cpu").manual_seed(77)
    ids = torch.stack([torch.zeros(b, dtype=torch.long), torch.arange(b)], 1).to(dev)
    return {"rgb": torch.rand(b, 3, 224, 224, generator=g).to(dev), "text": torch.randint(0, 30522, (b, 77), generator=g).to(dev),
            "audio": torch.randn(b, 1, 1001, 64, generator=g).to(dev), "example_ids": {"rgb": ids, "text": ids, "audio": ids}}


def _three_tower_stock(rgb, text, audio, b: int, dev, small: bool, steps: int, warmup: int):
    """Baseline leg of configs[3]: stock HF modules (SDPA, hipBLASLt, ATen elementwise), torch.optim.AdamW, F.normalize and the
    reference's loss op sequence over the three weighted pairs (oracle/eager_torch.py = contrastive.py:113-160), one stream."""
    from oracle.eager_torch import EagerContrastiveLoss

    import mmlearn_amd.tasks.contrastive_pretraining as cp
    from mmlearn_amd.tasks import ContrastivePretraining, LossPairSpec, ModuleKeySpec

    width = 128 if small else 768
    saved = cp.l2_normalize
    cp.l2_normalize = lambda x: torch.nn.functional.normalize(x, p=2, dim=-1)
    try:
        task = ContrastivePretraining(
            encoders={"rgb": rgb, "text": text, "audio": audio},
            heads={"shared": {"proj": nn.Linear(width, 512, bias=False)}},
            modality_module_mapping={m: ModuleKeySpec(encoder_key=m, head_key="shared") for m in ("rgb", "text", "audio")},
            loss=EagerContrastiveLoss(), optimizer=_adamw(False),
            modality_loss_pairs=[LossPairSpec(("rgb", "text")), LossPairSpec(("rgb", "audio"), 0.5), LossPairSpec(("text", "audio"), 0.5)],
            compute_validation_loss=False, compute_test_loss=False).to(dev)
        opt = task.configure_optimizers()
        batch = _three_tower_batch(b, dev)

        def step():
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = task.training_step(batch, 0)
            loss.backward()
            opt.step()
            return loss

        sec = _timed_steps(step, warmup, steps)
        loss = step()
        torch.cuda.synchronize()
    finally:
        cp.l2_normalize = saved
    return {"ms_per_step": round(sec * 1e3, 2), "samples_s": round(b / sec, 1), "steps": steps, "warmup": warmup, "per_gpu_batch": b,
            "loss": round(float(loss.detach().float()), 4),
            "what": "stock HF CLIP ViT-B/16 + BERT-base + HTSAT (CLAP audio), SDPA, hipBLASLt, torch.optim.AdamW, bf16 autocast, reference loss op sequence (eager), one stream"}


def ijepa_leg(b: int, dev, small: bool, steps: int = 8, warmup: int = 3, stock: bool = False):
    """BASELINE configs[4] on the driver's clock (bounded): I-JEPA ViT-L/16 224^2, 4 target blocks, EMA target encoder,
    12 x 384 predictor, bf16, AdamW + EMA update -- the step of tools/bench_ijepa_step.py (mmlearn/tasks/ijepa.py:217-263)."""
    from mmlearn_amd import _lib
    from tools.bench_ijepa_step import build

    task = build(small, not stock, dev)
    opt = task.configure_optimizers()
    opt = opt["optimizer"] if isinstance(opt, dict) else opt
    imgs = torch.rand(b, 3, 224, 224, generator=torch.Generator().manual_seed(1)).to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = task.training_step({"rgb": imgs}, 0)
        loss.backward()
        opt.step()
        task.on_before_zero_grad(opt)
        return loss

    torch.manual_seed(7)
    sec = _timed_steps(step, warmup + (2 if stock else 0), steps)   # MIOpen settles its patch-embedding convolution over the first calls
    if stock:
        loss = step()
        torch.cuda.synchronize()
        return {"ms_per_step": round(sec * 1e3, 2), "images_s": round(b / sec, 1), "steps": steps, "warmup": warmup + 2, "per_gpu_batch": b,
                "loss": round(float(loss.detach().float()), 4),
                "what": "the same ViT-L/16 + predictor as stock pre-LN blocks on SDPA / ATen LayerNorm / GELU, hipBLASLt, MIOpen patch convolution, "
                        "torch.optim.AdamW, bf16 autocast (tools/bench_ijepa_step.py build(fused=False)); mask generation, target / context "
                        "gathers, loss and EMA are this repo's ops in both legs"}
    _lib.profile_read()
    _lib.profile_enable(True)
    loss = step()
    torch.cuda.synchronize()
    prof = _lib.profile_read()
    _lib.profile_enable(False)
    roof = None
    dim = 256 if small else 1024
    if "ema_update" in prof:
        # dominant HBM-bound kernel of the I-JEPA-specific path: the EMA teacher update.  Algorithmic bytes:
        # copy mode reads the f32 student and writes the f32 teacher once = 8 B per parameter.
        n_param = sum(p.numel() for p in task.encoder.state_dict().values())
        cnt, ms = prof["ema_update"]
        gbs = 8.0 * n_param / (ms / cnt * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": "ema_update", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
                "traffic": None, "avg_launch_us": round(ms / cnt * 1e3, 2), "launches": cnt,
                "path_kernel_us": {k: round(v[1] / v[0] * 1e3, 2) for k, v in prof.items() if k in ("ema_update", "ijepa_loss_fwd", "ijepa_loss_bwd", "gather_rows", "scatter_rows", "pred_assemble", "pred_assemble_bwd")}}
    out = {"workload": f"BASELINE configs[4] on 1 GPU: I-JEPA ViT-L/16 224^2 (dim {dim}), 4 target blocks, EMA target encoder, 12x384 predictor, batch {b}, bf16",
           "ms_per_step": round(sec * 1e3, 2), "images_s": round(b / sec, 1), "steps": steps, "warmup": warmup, "loss": round(float(loss.detach().float()), 4),
           "roofline": roof}
    del task, opt, imgs
    torch.cuda.empty_cache()
    return out


def _loss_roofline(prof: dict, n_rows: int, n_cols: int, d: int, n_pairs: int, steps: int, traffic=None, loss_only: bool = False):
    """Roofline object of the loss path's MFMA kernels from HIP-event durations.  Algorithmic FLOPs: at one
    rank ONE [N, N, D] product per pair in the forward (2 N^2 D) and two products in the backward (dA = G B, dB = G^T A:
    4 N^2 D); row-sharded over W ranks 2 x 2 R C D forward and 2 x 2 R C D backward per rank.  Kernels that recompute the
    similarity tile count what they must produce, not what they execute."""
    single = n_rows == n_cols
    fwd = (2.0 if single else 4.0) * n_rows * n_cols * d * n_pairs
    bwd = 4.0 * n_rows * n_cols * d * n_pairs
    algo = {"clip_fwd": fwd, "clip_bwd": bwd, "sim_stats": fwd, "grad_gemm": bwd, "sim_grad": 0.0,
            # row-sharded directions: ONE launch recomputes the tiles and forms dX (csrc/clip_bwd.hip); it executes twice the
            # backward's algorithmic 4 R C D (the recompute), counted once
            "clip_bwd_fused": bwd,
            # one rank, <= 1024 matched rows per pair: ONE launch computes S, its statistics, G and both gradient products
            # (csrc/clip_fused.hip): all 6 N^2 D algorithmic FLOPs of the pair belong to it
            "clip_fused": fwd + bwd}
    if loss_only and prof.get("wgrad", (0, 0.0))[0] > 0:
        # large mirrored pair: dA = G B by the NT gradient GEMM, dB = G^T A by the transposed-read kernel, half of the
        # backward's algorithmic work each (in a whole-step profile "wgrad" is the encoders' weight gradient: not counted)
        algo["grad_gemm"] = bwd / 2
        algo["wgrad"] = bwd / 2
    mfma = {k: v for k, v in prof.items() if k in algo and v[0] > 0}
    if not mfma:
        return None
    dom = max(mfma, key=lambda k: mfma[k][1])
    rep = dom if algo[dom] > 0 else max((k for k in mfma if algo[k] > 0), key=lambda k: mfma[k][1])
    cnt, ms = mfma[rep]
    launches_per_step = cnt / max(steps, 1)
    avg_s = ms / cnt * 1e-3
    achieved = algo[rep] / launches_per_step / avg_s / 1e12
    if isinstance(traffic, dict):   # per-kernel HBM bytes per launch (PMC): quote the representative kernel's
        traffic = (traffic.get(rep) or {}).get("hbm_bytes_per_launch")
    return {"bound": "mfma", "kernel": rep, "achieved": round(achieved, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 5), "traffic": traffic, "avg_launch_us": round(avg_s * 1e6, 2), "launches": cnt,
            "dominant_by_time": dom, "shape": {"rows": n_rows, "cols": n_cols, "d": d, "pairs": n_pairs},
            "loss_path_kernel_us": {k: round(v[1] / v[0] * 1e3, 2) for k, v in prof.items() if v[0] > 0}}


def loss_n8192_leg(dev, n: int = 8192, d: int = 512, iters: int = 10):
    """The loss path alone at the global-batch size of BASELINE configs[2] (N = 8192, D = 512, bf16) on one GPU: where its
    MFMA kernels leave the launch-latency regime.  Bounded (about 30 launches), after the timed region."""
    from mmlearn_amd import ContrastiveLoss, LossPairSpec, _lib

    torch.manual_seed(0)
    a = torch.nn.functional.normalize(torch.randn(n, d, device=dev), dim=-1).bfloat16().requires_grad_(True)
    b = torch.nn.functional.normalize(torch.randn(n, d, device=dev), dim=-1).bfloat16().requires_grad_(True)
    ids = torch.stack([torch.zeros(n, dtype=torch.long, device=dev), torch.arange(n, device=dev)], 1)
    s = torch.tensor(1 / 0.07, device=dev, requires_grad=True)
    fn, pairs = ContrastiveLoss(), [LossPairSpec(("rgb", "text"))]

    def step(paired=None):
        a.grad = b.grad = s.grad = None
        loss = fn({"rgb_embedding": a, "text_embedding": b}, {"rgb": ids, "text": ids}, s, pairs, fully_paired=paired)
        loss.float().backward()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    _lib.profile_read()
    _lib.profile_enable(True)
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    prof = _lib.profile_read()
    _lib.profile_enable(False)
    # wall time from a pass of its own: the profiled pass of a fresh process creates every HIP event it records (hipEventCreate,
    # ~0.1 ms each), which made the r02 line read 4.9 ms of "wall" next to 0.5 ms of device time
    with _gc_quiet():
        t0 = time.perf_counter()
        for _ in range(iters):
            step()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / iters
        # the same with the collator's `fully_paired` hint (mmlearn_amd.wire): no matcher launch and, above all, no host read of its
        # result in the middle of the path -- in a training step that read is taken ahead of the encoders (prefetch_match); here,
        # with nothing in front of the loss, it leaves the device idle while the host enqueues the rest
        for _ in range(3):
            step(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            step(True)
        torch.cuda.synchronize()
        wall_paired = (time.perf_counter() - t0) / iters
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)))
        traffic = pmc.get("n8192", {}).get("per_kernel") or pmc.get("n8192", {}).get("hbm_bytes_per_launch")
    except Exception:
        traffic = None
    roof = _loss_roofline(prof, n, n, d, 1, iters, traffic, loss_only=True)
    if roof is not None:
        roof["traffic_from"] = f"profiles/{PMC_TRAFFIC_FILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of round 6, not of this run)"
        dev_us = sum(v[1] for k, v in prof.items()) / iters * 1e3
        roof["device_us_fwd_bwd"] = round(dev_us, 1)
        roof["wall_us_fwd_bwd"] = round(wall * 1e6, 1)
        roof["wall_us_fwd_bwd_paired_hint"] = round(wall_paired * 1e6, 1)
        roof["algorithmic_tflops_fwd_bwd"] = round(6.0 * n * n * d / (dev_us * 1e-6) / 1e12, 1)
    return roof


PMC_TRAFFIC_FILE = "r06_pmc_traffic.json"                # loss kernels at N = 1024 / 8192 (tools/probes/r6_pmc.sh)
PMC_TRAFFIC_SHARD_FILES = ("r06_pmc_traffic_shard.json", "r05_pmc_traffic_shard.json")   # C = 8192 from round 6; 2048 / 4096 from round 5


def _shard_traffic(cols: int):
    """HBM-side bytes of one rank's share at this column count from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE over tools/bench_loss_shard.py: round 6 for C = 8192, round 5 for the smaller widths), or None."""
    for name in PMC_TRAFFIC_SHARD_FILES:
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            continue
        if f"cols{cols}" in pmc:
            return dict(pmc[f"cols{cols}"], traffic_from=f"profiles/{name}")
    return None


def loss_shard_leg(dev, rows: int = 1024, cols: int = 8192, d: int = 512, rank: int = 3, iters: int = 20):
    """One rank's share of the row-sharded loss path at BASELINE configs[2] (W = 8: R = 1024 owned rows x C = 8192 gathered columns,
    both directions S_r = A_r B_all^T and T_r = B_r A_all^T) on one GPU: the launches ``mmlearn_amd.losses`` issues between the
    all-gather and the LSE all-reduce and after it (pack, similarity statistics, merge, gradient tiles, gradient GEMMs, finalize;
    the collectives themselves are not in it) -- the shape a rank runs at the BASELINE metric, on the driver's clock.  Algorithmic
    work per rank = 8 R C D."""
    from mmlearn_amd import _lib, kernels as K

    R, C, D = rows, cols, d
    p0 = min(rank, C // R - 1) * R
    torch.manual_seed(0)
    A = torch.nn.functional.normalize(torch.randn(C, D, device=dev), dim=-1).bfloat16()
    B = torch.nn.functional.normalize(torch.randn(C, D, device=dev), dim=-1).bfloat16()
    scale = torch.tensor([1 / 0.07], device=dev)
    upstream = torch.ones((), device=dev)
    comp = _lib.COMPUTE_BF16
    kg = 1.0 / (2.0 * C)

    want_t = not K.backward_recomputes_on_chip(R, C, D, comp, 2)   # as mmlearn_amd.losses decides: no transposed copies for the one-kernel backward

    def step():
        (ag, agt), (bg, bgt) = K.pack_rows_many([(A, None, C, False, want_t), (B, None, C, False, want_t)], comp)
        dirs = []
        for x, y, yt in ((K.slice_packed(ag, p0), bg, bgt), (K.slice_packed(bg, p0), ag, agt)):
            dirs.append(K.Direction(x=x, y=y, y_t=yt, r=R, c=C, label_off=p0, kappa=kg, ds_kappa=kg))
        dirs[1].s_row = dirs[1].s_col = dirs[1].s_diag = 0.0
        K.clip_forward(dirs, D, comp, scale)
        # stand-in for the all-reduced column LSEs: this rank's own rows are real, the other ranks' entries reuse them
        for dr, other in ((dirs[0], dirs[1]), (dirs[1], dirs[0])):
            dr.lse_col = other.lse.repeat(C // R).contiguous()
            dr.dx = torch.zeros((R, D), dtype=torch.bfloat16, device=dev)
        ds = torch.zeros(1, device=dev)
        K.clip_backward(dirs, D, comp, scale, upstream, ds)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    _lib.profile_read()
    _lib.profile_enable(True)
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    prof = _lib.profile_read()
    _lib.profile_enable(False)
    enq = []
    with _gc_quiet():
        t0 = time.perf_counter()
        for _ in range(iters):
            t1 = time.perf_counter()
            step()
            enq.append((time.perf_counter() - t1) * 1e6)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / iters
    pmc = _shard_traffic(C)
    roof = _loss_roofline(prof, R, C, D, 1, iters, pmc.get("per_kernel") if pmc else None, loss_only=True)
    if roof is not None:
        dev_us = sum(v[1] for v in prof.values()) / iters * 1e3
        roof["device_us_per_rank_share"] = round(dev_us, 1)
        roof["wall_us_per_rank_share"] = round(wall * 1e6, 1)
        roof["host_enqueue_us_per_rank_share"] = round(sorted(enq)[len(enq) // 2], 1)   # median: Python + ctypes + launches, no device wait
        roof["algorithmic_tflops_per_rank_share"] = round(8.0 * R * C * D / (dev_us * 1e-6) / 1e12, 1)
        roof["hbm_bytes_per_rank_share"] = pmc.get("total_hbm_bytes") if pmc else None
        roof["traffic_from"] = (pmc.get("traffic_from", "") + " (rocprofv3 --pmc passes, not of this run)") if pmc else None
        roof["operand_bytes_per_rank_share"] = int(2 * (C + R) * D * 2 * 2 + 2 * R * D * 2)
    return roof


def _leg_loss_shard(args, dev):
    return loss_shard_leg(dev)


def _leg_eager(args, dev):
    return eager_gpu_leg(args.batch, int(os.environ.get("RANK", "0")), dev, args.small)


def _leg_eager_tuned(args, dev):
    """The stock step once more with the library-GEMM selections file the HIP legs load (mmlearn_amd/tuned): the second denominator --
    `vs_baseline_tuned_eager` -- so that the ratio does not credit this package with what a TunableOp file gives stock PyTorch too."""
    tuned_gemms = _enable_tuned_gemms(args)
    out = eager_gpu_leg(args.batch, int(os.environ.get("RANK", "0")), dev, args.small)
    out["library_gemm_selection"] = "mmlearn_amd/tuned/gemm_gfx950.csv" if tuned_gemms else "library default"
    return out


def _leg_padded_text(args, dev):
    tuned_gemms = _enable_tuned_gemms(args)
    out = padded_text_leg(args.batch, dev, args.small)
    out["library_gemm_selection"] = "mmlearn_amd/tuned/gemm_gfx950.csv" if tuned_gemms else "library default"
    return out


def _leg_padded_text_eager(args, dev):
    return eager_gpu_leg(args.batch, 0, dev, args.small, padded=True)


def _leg_loss_n8192(args, dev):
    return loss_n8192_leg(dev)


def _enable_tuned_gemms(args) -> bool:
    """The product's library-GEMM selections (mmlearn_amd/tuned: TunableOp look-up, no tuning at run time) for the HIP legs;
    the stock-step leg never calls this.  ``MMK_BENCH_NO_TUNED=1`` keeps the library's default heuristic (A/B switch)."""
    if args.small or args.no_fused_encoder_ops or os.environ.get("MMK_BENCH_NO_TUNED"):
        return False
    from mmlearn_amd import tuned

    ok = tuned.enable()
    if not ok:
        print("[bench] tuned GEMM selections refused (another PyTorch / hipBLASLt / rocBLAS build?): library defaults", file=sys.stderr)
    return ok


def _leg_three_tower(args, dev):
    tuned_gemms = _enable_tuned_gemms(args)
    out = three_tower_leg(64 if args.small else 256, dev, args.small)
    out["library_gemm_selection"] = "mmlearn_amd/tuned/gemm_gfx950.csv" if tuned_gemms else "library default"
    return out


def _leg_full_last_layer(args, dev):
    tuned_gemms = _enable_tuned_gemms(args)
    out = full_last_layer_leg(args.batch, dev, args.small)
    out["library_gemm_selection"] = "mmlearn_amd/tuned/gemm_gfx950.csv" if tuned_gemms else "library default"
    return out


def _leg_ijepa(args, dev):
    tuned_gemms = _enable_tuned_gemms(args)   # the product's library-GEMM selections (look-up only); the stock leg never loads them
    out = ijepa_leg(16 if args.small else 128, dev, args.small)
    out["library_gemm_selection"] = "mmlearn_amd/tuned/gemm_gfx950.csv" if tuned_gemms else "library default"
    return out


def _leg_three_tower_eager(args, dev):
    return three_tower_leg(64 if args.small else 256, dev, args.small, stock=True)


def _leg_ijepa_eager(args, dev):
    return ijepa_leg(16 if args.small else 128, dev, args.small, stock=True)


LEGS = {"eager_gpu": _leg_eager, "eager_gpu_tuned": _leg_eager_tuned, "padded_text": _leg_padded_text, "padded_text_eager": _leg_padded_text_eager,
        "loss_n8192": _leg_loss_n8192, "loss_shard": _leg_loss_shard, "three_tower": _leg_three_tower, "ijepa_vitl": _leg_ijepa,
        "full_last_layer": _leg_full_last_layer, "three_tower_eager": _leg_three_tower_eager, "ijepa_vitl_eager": _leg_ijepa_eager}
LEG_TIMEOUT_S = {"eager_gpu": 240, "eager_gpu_tuned": 240, "padded_text": 150, "padded_text_eager": 240, "loss_n8192": 120, "loss_shard": 120, "three_tower": 150, "ijepa_vitl": 240, "full_last_layer": 150,
                 "three_tower_eager": 180, "ijepa_vitl_eager": 240}


def leg_main(args) -> int:
    """``python bench.py --leg NAME``: one bounded leg on cuda:LOCAL_RANK, its JSON object as the last stdout line."""
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import faulthandler

    # a leg that stops making progress says where (all Python threads) shortly before the parent gives up on it
    faulthandler.dump_traceback_later(max(LEG_TIMEOUT_S[args.leg] - 20, 20), exit=False)
    if not torch.cuda.is_available():
        raise SystemExit("[bench] no GPU visible; the legs have no CPU path")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from mmlearn_amd import _lib

    _lib.check(_lib.lib().mmk_device_check())
    out = LEGS[args.leg](args, dev)
    torch.cuda.synchronize()
    print(json.dumps(out), flush=True)
    return 0


def run_leg(name: str, args) -> dict:
    """Start ``bench.py --leg name`` as a CHILD process (never an exec of this one: it has initialised the GPU) and read the
    JSON object it prints.  A leg that fails -- a Python error, a GPU fault that aborts the child, the time bound -- becomes
    ``{"error": ...}`` in the line; the headline numbers are already taken by then."""
    import subprocess

    if os.environ.get("MMK_BENCH_LEGS_INPROCESS") is not None:   # debugging switch: the leg in this very process
        print(f"[bench] leg {name} (in process) ...", file=sys.stderr, flush=True)
        return LEGS[name](args, torch.device("cuda", torch.cuda.current_device()))
    cmd = [sys.executable, os.path.abspath(__file__), "--leg", name, "--batch", str(args.batch)] + (["--small"] if args.small else [])
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "MASTER_ADDR", "MASTER_PORT", "MMK_BENCH_FORCE_DIST")}
    print(f"[bench] leg {name} ...", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    try:
        cp = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=LEG_TIMEOUT_S[name])
    except subprocess.TimeoutExpired as e:
        err = e.stderr.decode(errors="replace") if isinstance(e.stderr, bytes) else (e.stderr or "")
        print(f"[bench] leg {name}: no result within {LEG_TIMEOUT_S[name]} s; its stderr ended with:\n{err[-3000:]}", file=sys.stderr, flush=True)
        return {"error": f"timed out after {LEG_TIMEOUT_S[name]} s"}
    lines = [ln for ln in cp.stdout.splitlines() if ln.strip()]
    if cp.returncode == 0 and lines:
        try:
            out = json.loads(lines[-1])
            print(f"[bench] leg {name}: done in {time.perf_counter() - t0:.1f} s", file=sys.stderr, flush=True)
            return out
        except ValueError:
            pass
    tail = " | ".join(cp.stderr.strip().splitlines()[-3:])[-400:]
    print(f"[bench] leg {name}: exit code {cp.returncode}: {tail}", file=sys.stderr, flush=True)
    return {"error": f"exit code {cp.returncode}: {tail}"}


def _free_port() -> int:
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n: int, argv: list[str]) -> int:
    """``python bench.py --gpus N`` without a launcher: start N fresh rank processes (one per GPU) and relay rank 0's result
    line.  This parent never makes a HIP call -- the ranks are children started with ``subprocess`` (no re-exec of a process
    that has initialised the GPU) -- and exits non-zero as soon as any rank does (the others are then ended by PID)."""
    import subprocess

    env = dict(os.environ)
    env.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        # rank 0's stdout is captured (its last line is the result); the other ranks' stdout goes to our stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0) or None))
    out0 = ""
    rc = 0
    try:
        import threading

        def pump():
            nonlocal out0
            out0 = procs[0].stdout.read()

        th = threading.Thread(target=pump, daemon=True)
        th.start()
        live = set(range(n))
        while live:
            for r in list(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print(f"[bench] rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr)
                    for q in live:
                        procs[q].terminate()
            time.sleep(0.05)
        th.join(timeout=10)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    lines = [ln for ln in out0.splitlines() if ln.strip()]
    for ln in lines[:-1]:
        print(ln, file=sys.stderr)
    if rc == 0 and lines:
        print(lines[-1], flush=True)    # the JSON line: last line on stdout
    elif lines:
        print(lines[-1], file=sys.stderr)
    return rc


def main():
    # HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  At N > 1 the step has the two towers' streams
    # plus RCCL's, and with 4 queues two of them can land on one queue: the towers' overlap was then lost (216 vs 207.6 ms in
    # the 1-rank dry run).  Must be set before the first HIP call of the process.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="per-GPU batch (BASELINE: 1024)")
    ap.add_argument("--small", action="store_true", help="tiny encoders (debug only; invalid as a result)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-eager-leg", action="store_true", help="skip the stock-step leg (vs_baseline becomes null)")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the bounded configs[3] / configs[4] / N=8192 legs")
    ap.add_argument("--no-fused-encoder-ops", action="store_true", help="keep torch LayerNorm / HF quick-GELU in the encoders")
    ap.add_argument("--leg", choices=sorted(LEGS), default=None,
                    help="run ONE of the bounded legs alone and print its JSON object (how the headline run starts them: each in a "
                         "process of its own, so that a leg can fail without taking the headline line with it)")
    args = ap.parse_args()
    if args.leg is not None:
        raise SystemExit(leg_main(args))

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare ``python bench.py --gpus N``: be the launcher (the torchrun form keeps working: it sets WORLD_SIZE)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus must agree")
    if not torch.cuda.is_available():
        raise SystemExit(f"[bench] rank {rank}: no GPU visible (torch.cuda.is_available() is False); this benchmark has no CPU path")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"[bench] rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # MMK_BENCH_FORCE_DIST=1: run the N > 1 code path (RCCL process group, DDP, gathered negatives) on a 1-rank group --
    # a dry run of the multi-GPU bench on a single-GPU box; the JSON line says so in config.parallelism
    force_dist = world == 1 and os.environ.get("MMK_BENCH_FORCE_DIST") is not None
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29631")
        dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    from mmlearn_amd import ContrastiveLoss, _lib

    _lib.check(_lib.lib().mmk_device_check())
    tuned_gemms = _enable_tuned_gemms(args)
    loss_fn = ContrastiveLoss(static_shapes=True)
    loss_fn._force_gather = force_dist
    task = build_task(loss_fn, args.small, fused=not args.no_fused_encoder_ops).to(dev)
    opt = task.configure_optimizers()
    stepper = _Step(task)
    if world > 1 or force_dist:
        if getattr(task, "concurrent_encoders", False) and not os.environ.get("MMK_BENCH_OUTER_DDP"):
            # one DDP instance per tower, each built under its tower's stream: gradient accumulation, buckets and all-reduces
            # stay on that stream and the towers' backward passes keep overlapping (a single outer DDP serialises them)
            # buffers of these towers are constants (position ids): no per-forward broadcast
            task.wrap_towers_in_ddp(broadcast_buffers=False)
        else:
            stepper = nn.parallel.DistributedDataParallel(stepper, device_ids=[local_rank], gradient_as_bucket_view=True)
    batch = synthetic_batch(args.batch, rank, dev)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = stepper(batch)
        loss.backward()
        opt.step()
        return loss

    def fence():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # Per-kernel durations come from HIP events stamped around every launch (csrc/runtime.hip).  With the towers on separate
    # streams (the default) a kernel's own duration is not observable in the timed region -- overlapped kernels share the
    # CUs, and event packets on three busy queues add 4-6 us per bracket -- so the events are then taken in a follow-up
    # pass of the same step on one stream, right after the timed region; with MMK_BENCH_NO_STREAMS=1 they are taken in it.
    overlapped = bool(getattr(task, "concurrent_encoders", False))
    if not overlapped:
        _lib.profile_read()
        _lib.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    prof_steps = args.steps
    if overlapped:
        task.concurrent_encoders, task.match_ahead = False, False
        step()
        fence()
        _lib.profile_read()
        _lib.profile_enable(True)
        prof_steps = max(2, min(args.steps, 4))
        for _ in range(prof_steps):
            step()
        fence()
        task.concurrent_encoders, task.match_ahead = True, True
    prof = _lib.profile_read()
    _lib.profile_enable(False)
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1 or force_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = t.item()
    final_loss = float(loss.detach().float().item())
    if rank == 0:
        print(f"[bench] peak HBM allocated {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", file=sys.stderr)

    # ---- legs outside the headline timed region (bounded; all on the driver's clock) ------------------------------
    del stepper, opt, task, batch, loss, loss_fn
    torch.cuda.empty_cache()
    # Each leg runs in a child process of its own (run_leg): this process keeps its result whatever happens there.
    eager = None
    if not args.no_eager_leg and not args.no_fused_encoder_ops:
        # every rank times the stock step on its own GPU (no collectives); rank 0's figure is reported
        eager = run_leg("eager_gpu", args)
    extra = {}
    if rank == 0 and world == 1 and not force_dist and not args.no_extra_legs:
        for name in ("loss_n8192", "loss_shard", "three_tower", "ijepa_vitl", "full_last_layer", "padded_text", "eager_gpu_tuned"):
            extra[name] = run_leg(name, args)
        if isinstance(extra.get("padded_text"), dict) and "pairs_s" in extra["padded_text"]:
            base = run_leg("padded_text_eager", args)
            extra["padded_text"]["eager"] = base
            if "pairs_s" in base:
                extra["padded_text"]["vs_baseline"] = round(extra["padded_text"]["pairs_s"] / base["pairs_s"], 3)
        # denominators of the configs[3] / configs[4] legs: the same steps on stock modules, each in a child process of its own
        for name, key in (("three_tower", "samples_s"), ("ijepa_vitl", "images_s")):
            if isinstance(extra.get(name), dict) and "error" not in extra[name]:
                base = run_leg(name + "_eager", args)
                extra[name]["eager"] = base
                if key in base and key in extra[name]:
                    extra[name]["vs_baseline"] = round(extra[name][key] / base[key], 3)

    if rank == 0 and eager and "pairs_s" in eager and isinstance(extra.get("full_last_layer"), dict) and "pairs_s" in extra["full_last_layer"]:
        extra["full_last_layer"]["vs_baseline"] = round(extra["full_last_layer"]["pairs_s"] / eager["pairs_s"], 3)
    if rank == 0:
        n_rows, n_cols, d = args.batch, args.batch * world, 512
        traffic = None  # HBM bytes per launch from rocprofv3 PMC passes of the same kernel and shape (profiles/)
        try:
            if world == 1 and args.batch == 1024:   # (also the 1-rank dry run of the N > 1 path: one rank still runs the one-launch kernel)
                pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)))
                traffic = pmc.get("n1024", {}).get("hbm_bytes_per_launch")
            elif args.batch == 1024:   # N > 1: the rank's sharded shape (R = batch rows x C = batch * world columns)
                traffic = (_shard_traffic(n_cols) or {}).get("per_kernel")
        except Exception:
            traffic = None
        roofline = _loss_roofline(prof, n_rows, n_cols, d, 1, prof_steps, traffic)
        if roofline is not None:
            roofline["events_from"] = "single-stream pass after the timed region" if overlapped else "timed region"
            roofline["traffic_from"] = (f"profiles/{PMC_TRAFFIC_FILE if world == 1 else PMC_TRAFFIC_SHARD_FILES[0]} (rocprofv3 --pmc FETCH_SIZE / "
                                        "WRITE_SIZE passes of round 6 over the same kernel and shape, not of this run)") if traffic is not None else None
        # the dominant hand-written kernel of the whole step (SURVEY 8(f1) widening): the weight-gradient GEMM.  Algorithmic
        # FLOPs = 2 M N K summed over the encoder Linears it serves (HISTORY.md 5.5), time from the same HIP-stamped events.
        roofline_widened = None
        if "wgrad" in prof and not args.small and not args.no_fused_encoder_ops:
            m_v, m_t, e = args.batch * 197, args.batch * 77, 768
            per_step = (12 * 2.0 * m_v * e * (3 * e + e + 4 * e + 4 * e) + 2.0 * args.batch * 196 * e * e
                        + 12 * 2.0 * m_t * e * (3 * e + e + 4 * e + 4 * e))
            cnt, ms = prof["wgrad"]
            achieved = per_step * prof_steps / (ms * 1e-3) / 1e12
            roofline_widened = {"bound": "mfma", "kernel": "wgrad", "achieved": round(achieved, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                                "unit": "TFLOP/s", "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None,
                                "avg_launch_us": round(ms / cnt * 1e3, 1), "launches": cnt,
                                "share_of_step_time": round(ms / 1e3 / prof_steps / (dt / args.steps), 3)}
        out = {
            "metric": "image-text pairs/s (whole node), ViT-B/16+BERT-base contrastive step",
            "value": round(args.batch * world * args.steps / dt, 2),
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            # value / (N x the stock step's pairs/s on one GPU): the stock step is timed at N = 1 shape (local negatives, no
            # gradient all-reduce), i.e. the baseline is credited with perfect scaling
            "vs_baseline": round(args.batch * world * args.steps / dt / (eager["pairs_s"] * world), 3) if eager and "pairs_s" in eager else None,
            # the same ratio against the stock step run WITH this package's library-GEMM selections file (the eager_gpu_tuned leg)
            "vs_baseline_tuned_eager": (round(args.batch * world * args.steps / dt / (extra["eager_gpu_tuned"]["pairs_s"] * world), 3)
                                        if isinstance(extra.get("eager_gpu_tuned"), dict) and "pairs_s" in extra["eager_gpu_tuned"] else None),
            # like-for-like with rounds <= 4: the same step with the last layer of both towers computed for ALL tokens
            "full_last_layer": ({k: extra["full_last_layer"].get(k) for k in ("ms_per_step", "pairs_s", "vs_baseline")}
                                if isinstance(extra.get("full_last_layer"), dict) and "pairs_s" in extra["full_last_layer"] else None),
            "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: CLIP ViT-B/16 + BERT-base, D=512 projection, bf16 autocast, "
                                   f"per-GPU batch {args.batch}, {'local' if world == 1 else 'global-batch (all-gather)'} negatives"
                                   + (" [DEBUG small encoders]" if args.small else ""),
                       "global_batch": args.batch * world, "parallelism": f"dp{world}" + (" (1-rank RCCL dry run of the N > 1 path)" if force_dist else ""), "loss": "mmlearn_amd.ContrastiveLoss (HIP)",
                       "library_gemm_selection": "mmlearn_amd/tuned/gemm_gfx950.csv (TunableOp look-up, no tuning at run time)" if tuned_gemms else "library default",
                       "encoder_ops": "torch" if args.no_fused_encoder_ops else "HIP LayerNorm (+ fused residual add / dropout / deferred biases), MLP GEMMs with the activation in the epilogue, fused-QKV attention, weight-gradient GEMM, last layer of each tower for token 0 only (both towers pool token 0: accelerate_encoder's cls_only='auto' proved it; exact) (mmlearn_amd.fused / .attention)",
                       "final_loss": round(final_loss, 4),
                       "final_loss_note": "random-init towers on random pixels / tokens emit near-identical embeddings, so the loss sits at ln(batch); "
                                          "the timed work does not depend on the values (numerics are covered by tests/, not by this line)"},
            "roofline": roofline,
            "roofline_widened": roofline_widened,
            "eager_gpu": eager,
            "roofline_n8192": extra.get("loss_n8192"),
            # one rank's share of configs[2] (R = 1024 x C = 8192, both directions): the shape a rank runs at the BASELINE metric
            "roofline_shard": extra.get("loss_shard"),
            "extra": {k: v for k, v in extra.items() if k not in ("loss_n8192", "loss_shard")} or None,
        }
        if not args.no_cpu_baseline:
            # rank 0's host cores, at every N (the other ranks wait at the barrier below): a bounded sample of the same step
            out["cpu_baseline"] = cpu_baseline()
    # RCCL leaves its version banner in the C stdout buffer of a process until exit: every rank pushes its buffer out, then all
    # meet, and only then rank 0 prints the result line -- the last line on stdout
    try:
        import ctypes

        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if world > 1 or force_dist:
        dist.barrier()
        torch.cuda.synchronize()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
