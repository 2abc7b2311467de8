"""BASELINE configs[0] on the GPU (2-layer MLP encoders, two modalities, batch 64): the whole training step -- encoders, HIP
l2-normalise, the one-launch loss with its gradients, backward, AdamW -- captured ONCE into a HIP graph (`torch.cuda.CUDAGraph`)
and replayed, next to the same step launched eagerly.  SURVEY 8(f1) lists graph capture of the step; at the headline batch
(B = 1024, 200 ms of device work queued in 30 ms) it buys nothing, at this size the step is all launch overhead.

    python tools/graph_step.py [--batch 64] [--dim 512] [--iters 200]

Prints one JSON line: wall per step eager / graph, and the largest difference between the two after the same number of steps
(same seeds: the captured step must BE the eager step)."""
import argparse
import copy
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
import torch.nn as nn


class _MLP(nn.Module):
    def __init__(self, key, d_in, d_hidden, d_out):
        super().__init__()
        self.key = key
        self.net = nn.Sequential(nn.Flatten(1), nn.Linear(d_in, d_hidden), nn.GELU(), nn.Linear(d_hidden, d_out))

    def forward(self, inputs):
        return (self.net(inputs[self.key]),)


def build(dev, dim, own_adamw=False):
    from functools import partial

    from mmlearn_amd import ContrastiveLoss
    from mmlearn_amd.optim import AdamW as OwnAdamW
    from mmlearn_amd.tasks import ContrastivePretraining, LossPairSpec

    torch.manual_seed(0)
    task = ContrastivePretraining(
        encoders={"rgb": _MLP("rgb", 3 * 16 * 16, 1024, dim), "text": _MLP("text", 77, 1024, dim)},
        loss=ContrastiveLoss(), optimizer=partial(OwnAdamW if own_adamw else torch.optim.AdamW, lr=1e-3, capturable=True),
        modality_loss_pairs=[LossPairSpec(("rgb", "text"))], compute_validation_loss=False, compute_test_loss=False).to(dev)
    return task


def make(dev, dim, own_adamw=False):
    task = build(dev, dim, own_adamw)
    opt = task.configure_optimizers()
    return task, (opt["optimizer"] if isinstance(opt, dict) else opt)


def make_batch(b, dev, shuffled=False, seed=3):
    """``shuffled``: the text rows arrive in another order than the images (ids permuted, no ``fully_paired`` promise), so the
    loss has to run the id matcher."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.stack([torch.zeros(b, dtype=torch.long), torch.arange(b)], 1).to(dev)
    batch = {"rgb": torch.rand(b, 3, 16, 16, generator=g).to(dev), "text": torch.rand(b, 77, generator=g).to(dev),
             "example_ids": {"rgb": ids, "text": ids}, "fully_paired": True}
    if shuffled:
        perm = torch.randperm(b, generator=g).to(dev)
        batch["example_ids"] = {"rgb": ids, "text": ids[perm]}
        del batch["fully_paired"]
    return batch


def step(task, opt, batch):
    opt.zero_grad(set_to_none=False)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = task.training_step(batch, 0)
    loss.backward()
    opt.step()
    return loss


def ijepa_main(a):
    """--ijepa: the I-JEPA ViT-S/16 step (EMA teacher, 6 x 384 predictor), eager vs ``mmlearn_amd.graph.CapturedIJEPAStep``."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import bench_ijepa_step as T
    from mmlearn_amd.graph import CapturedIJEPAStep

    dev = torch.device("cuda", 0)
    imgs = torch.rand(a.batch, 3, 224, 224, generator=torch.Generator().manual_seed(1)).to(dev)
    out = {"workload": f"I-JEPA ViT-S/16 + 6x384 predictor, batch {a.batch}, bf16 autocast, own AdamW(capturable), EMA teacher"}

    def make():
        task = T.build("vits", True, dev, capturable=True)
        opt = task.configure_optimizers()
        return task, (opt["optimizer"] if isinstance(opt, dict) else opt)

    task_e, opt_e = make()

    def eager():
        opt_e.zero_grad(set_to_none=False)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = task_e.training_step({"rgb": imgs}, 0)
        loss.backward()
        opt_e.step()
        task_e.on_before_zero_grad(opt_e)

    torch.manual_seed(3)
    for _ in range(5):
        eager()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        eager()
    torch.cuda.synchronize()
    out["eager_ms"] = round((time.perf_counter() - t0) / a.iters * 1e3, 3)
    task_g, opt_g = make()
    runner = CapturedIJEPAStep(task_g, opt_g, warmup=3)
    torch.manual_seed(3)
    for _ in range(3 + 40):    # warm-up + enough steps to have captured the common mask geometries
        runner({"rgb": imgs})
    torch.cuda.synchronize()
    n_before = len(runner.graphs)
    t0 = time.perf_counter()
    for _ in range(a.iters):
        runner({"rgb": imgs})
    torch.cuda.synchronize()
    out["graph_ms"] = round((time.perf_counter() - t0) / a.iters * 1e3, 3)
    out["graphs"] = {"before_timing": n_before, "after": len(runner.graphs), "keys": sorted(map(list, runner.graphs))}
    out["speedup"] = round(out["eager_ms"] / out["graph_ms"], 3)
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--ijepa", action="store_true", help="the I-JEPA ViT-S step instead of the configs[0] contrastive step")
    a = ap.parse_args()
    if a.ijepa:
        return ijepa_main(a)
    dev = torch.device("cuda", 0)
    b = a.batch
    batch = make_batch(b, dev)

    # ---- eager
    task_e, opt_e = make(dev, a.dim)
    for _ in range(5):
        step(task_e, opt_e, batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        loss_e = step(task_e, opt_e, batch)
    torch.cuda.synchronize()
    eager_us = (time.perf_counter() - t0) / a.iters * 1e6

    # ---- graph: the same 5 warm-up steps eagerly (on a side stream, as torch asks), then capture one step and replay it
    task_g, opt_g = make(dev, a.dim)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(5):
            step(task_g, opt_g, batch)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loss_g = step(task_g, opt_g, batch)
    torch.cuda.synchronize()
    # the capture itself did not run the step: a.iters replays = a.iters steps
    t0 = time.perf_counter()
    for _ in range(a.iters):
        graph.replay()
    torch.cuda.synchronize()
    graph_us = (time.perf_counter() - t0) / a.iters * 1e6

    diff = max((pe.detach().float() - pg.detach().float()).abs().max().item()
               for pe, pg in zip(task_e.parameters(), task_g.parameters()))
    scale = max(p.detach().float().abs().max().item() for p in task_e.parameters())
    print(json.dumps({"workload": f"configs[0] on the GPU: two 2-layer MLP encoders, batch {b}, D = {a.dim}, bf16 autocast, fully paired",
                      "eager_us_per_step": round(eager_us, 1), "graph_us_per_step": round(graph_us, 1),
                      "speedup": round(eager_us / graph_us, 2), "steps_each": 5 + a.iters,
                      "loss_eager": round(float(loss_e.detach()), 6), "loss_graph": round(float(loss_g.detach()), 6),
                      "max_param_diff_after_all_steps": diff, "param_scale": scale}))


if __name__ == "__main__":
    main()
