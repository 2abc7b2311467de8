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
                  sequence) on the host cores for a bounded sample (rank 0, N = 1 only).
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


class VisionEncoder(nn.Module):
    """HF CLIP ViT-B/16 with projection; mmlearn encoder contract: forward(dict) -> (embedding,)"""

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
        h = self.model(input_ids=inputs["text"]).last_hidden_state[:, 0]
        return (self.proj(h),)


class _Step(nn.Module):
    """DDP wraps forward(); route it to training_step like Lightning's strategy wrapper does."""

    def __init__(self, task):
        super().__init__()
        self.task = task

    def forward(self, batch):
        return self.task.training_step(batch, 0)


def synthetic_batch(b: int, rank: int, device):
    g = torch.Generator(device="cpu").manual_seed(1000 + rank)
    ids = torch.stack([torch.zeros(b, dtype=torch.long), torch.arange(rank * b, (rank + 1) * b)], 1)
    batch = {
        "rgb": torch.rand(b, 3, 224, 224, generator=g).to(device),
        "text": torch.randint(0, 30522, (b, 77), generator=g).to(device),
        "example_ids": {"rgb": ids.to(device), "text": ids.to(device)},
    }
    if os.environ.get("MMK_BENCH_PAIRED_HINT"):  # A/B: what mmlearn_amd.wire.DefaultDataCollator adds (default: the matcher runs)
        batch["fully_paired"] = True
    return batch


def _adamw(fused: bool):
    """AdamW(lr=1e-4, weight_decay=0.1): torch's, or the same update as one HIP launch per group (mmlearn_amd.optim)."""
    if fused and os.environ.get("MMK_BENCH_TORCH_ADAMW") is None:
        from mmlearn_amd.optim import AdamW

        return partial(AdamW, lr=1e-4, weight_decay=0.1)
    return partial(torch.optim.AdamW, lr=1e-4, weight_decay=0.1)


def build_task(loss, small: bool, fused: bool = False):
    from mmlearn_amd.tasks.contrastive_pretraining import ContrastivePretraining

    torch.manual_seed(0)
    rgb, text = VisionEncoder(small, hip_attention=fused), TextEncoder(small, hip_attention=fused)
    if fused:  # HIP LayerNorm / quick-GELU inside the encoders (same parameters, same math)
        from mmlearn_amd.fused import accelerate_encoder

        # CLIP is pre-LN: these LayerNorms feed autocast Linears only, so they may emit bf16 directly;
        # BERT is post-LN (the LN output is the residual stream) and keeps f32 outputs.
        add_ln = os.environ.get("MMK_BENCH_NO_ADD_LN") is None   # A/B switch for the fused residual add + LayerNorm
        accelerate_encoder(rgb, low_precision_ln=("layer_norm1", "layer_norm2", "post_layernorm"), fuse_qkv=True, fuse_add_ln=add_ln)
        accelerate_encoder(text, fuse_qkv=True, fuse_add_ln=add_ln)
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
    sequence (oracle/eager_torch.py).  Bounded sample: batches of 8 pairs until the time budget is spent (at most 32 threads)."""
    from oracle.eager_torch import EagerContrastiveLoss

    import mmlearn_amd.tasks.contrastive_pretraining as cp

    # many-core hosts: torch eager on ~200 threads is slower than on 32 (measured on the 256-core MI355X box:
    # 16 pairs took ~7 min at 256 threads); use at most 32 and report what was used
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    b = 8
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
    ap.add_argument("--no-fused-encoder-ops", action="store_true", help="keep torch LayerNorm / HF quick-GELU in the encoders")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
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
    loss_fn = ContrastiveLoss(static_shapes=True)
    loss_fn._force_gather = force_dist
    task = build_task(loss_fn, args.small, fused=not args.no_fused_encoder_ops).to(dev)
    opt = task.configure_optimizers()
    stepper = _Step(task)
    if world > 1 or force_dist:
        if getattr(task, "concurrent_encoders", False) and not os.environ.get("MMK_BENCH_OUTER_DDP"):
            # one DDP instance per tower, each built under its tower's stream: gradient accumulation, buckets and all-reduces
            # stay on that stream and the towers' backward passes keep overlapping (a single outer DDP serialises them)
            task.wrap_towers_in_ddp()
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

    if rank == 0:
        n_rows, n_cols, d = args.batch, args.batch * world, 512
        # algorithmic FLOPs per launch (DESIGN.md "Kernels"): every launch covers both directions of the pair
        algo = {"sim_stats": 2 * 2.0 * n_rows * n_cols * d,   # S_r and T_r row blocks (8RCD/4 per GEMM, two of them)
                "grad_gemm": 2 * 2.0 * n_rows * n_cols * d,   # dA_r = G_r B_all, dB_r = H_r A_all
                "sim_grad": 0.0}                               # tile recompute: implementation cost, not counted
        if world == 1:
            algo["sim_stats"] = 2.0 * n_rows * n_cols * d      # one [N,N,D] product is algorithmically enough
        mfma = {k: v for k, v in prof.items() if k in algo}
        dom = max(mfma, key=lambda k: mfma[k][1]) if mfma else None
        roofline = None
        if dom is not None:
            # report the dominant kernel with counted work; the recompute kernel has no algorithmic FLOPs of its own
            rep = dom if algo[dom] > 0 else max((k for k in mfma if algo[k] > 0), key=lambda k: mfma[k][1])
            cnt, ms = mfma[rep]
            avg_s = ms / cnt * 1e-3
            achieved = algo[rep] / avg_s / 1e12
            traffic = None  # HBM bytes per launch from rocprofv3 PMC passes of the same kernel and shape (profiles/)
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
                if world == 1 and args.batch == 1024:
                    traffic = pmc["n1024_tile64"][rep]["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
            roofline = {"bound": "mfma", "kernel": rep, "achieved": round(achieved, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 5), "traffic": traffic, "avg_launch_us": round(avg_s * 1e6, 2),
                        "launches": cnt, "dominant_by_time": dom,
                        "events_from": "single-stream pass after the timed region" if overlapped else "timed region",
                        "loss_path_kernel_us": {k: round(v[1] / v[0] * 1e3, 2) for k, v in prof.items()}}
        # the dominant hand-written kernel of the whole step (SURVEY 8(f1) widening): the weight-gradient GEMM.  Algorithmic
        # FLOPs = 2 M N K summed over the encoder Linears it serves, time from the same HIP-stamped events.
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
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: CLIP ViT-B/16 + BERT-base, D=512 projection, bf16 autocast, "
                                   f"per-GPU batch {args.batch}, {'local' if world == 1 else 'global-batch (all-gather)'} negatives"
                                   + (" [DEBUG small encoders]" if args.small else ""),
                       "global_batch": args.batch * world, "parallelism": f"dp{world}" + (" (1-rank RCCL dry run of the N > 1 path)" if force_dist else ""), "loss": "mmlearn_amd.ContrastiveLoss (HIP)",
                       "encoder_ops": "torch" if args.no_fused_encoder_ops else "HIP LayerNorm (+ fused residual add / dropout / deferred biases), bias+activation, fused-QKV attention, weight-gradient GEMM (mmlearn_amd.fused / .attention)",
                       "final_loss": round(final_loss, 4)},
            "roofline": roofline,
            "roofline_widened": roofline_widened,
        }
        if world == 1 and not args.no_cpu_baseline:
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
