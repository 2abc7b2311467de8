"""GPU: the HIP loss path timed against the reference's eager op sequence (oracle/eager_torch.py) on the same GPU.
Not a pass/fail performance gate; the measured numbers are printed and written to
gpurun_out/perf_vs_eager.json so they can be quoted in DESIGN.md."""

import json
import os
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


def _time(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


@pytest.mark.parametrize("n", [1024, 8192])
def test_loss_path_vs_reference_eager_sequence(n):
    from mmlearn_amd import ContrastiveLoss, LossPairSpec
    from oracle.eager_torch import EagerContrastiveLoss

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    d = 512
    a = torch.nn.functional.normalize(torch.randn(n, d, device=dev), dim=-1).requires_grad_(True)
    b = torch.nn.functional.normalize(torch.randn(n, d, device=dev), dim=-1).requires_grad_(True)
    ids = torch.stack([torch.zeros(n, dtype=torch.long, device=dev), torch.arange(n, device=dev)], 1)
    s = torch.tensor(1 / 0.07, device=dev, requires_grad=True)
    pairs = [LossPairSpec(("rgb", "text"))]
    res = {}
    for name, fn in (("hip", ContrastiveLoss()), ("eager", EagerContrastiveLoss())):
        def step():
            a.grad = b.grad = s.grad = None
            with torch.autocast("cuda", dtype=torch.bfloat16):   # Lightning bf16-mixed, like the reference run
                loss = fn({"rgb_embedding": a, "text_embedding": b}, {"rgb": ids, "text": ids}, s, pairs)
            loss.backward()
            return loss

        res[name] = {"ms": _time(step, 20 if n <= 1024 else 5), "loss": float(step().detach())}
    speedup = res["eager"]["ms"] / res["hip"]["ms"]
    rec = {"n": n, "d": d, "hip_ms": round(res["hip"]["ms"], 3), "eager_ms": round(res["eager"]["ms"], 3), "speedup": round(speedup, 2),
           "loss_hip": res["hip"]["loss"], "loss_eager": res["eager"]["loss"]}
    print(json.dumps(rec))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/perf_vs_eager.json", "a") as f:
        f.write(json.dumps(rec) + "\n")
    # same loss within the bf16 tolerance (the eager path rounds logits to bf16, the HIP path keeps f32 accumulators)
    assert abs(res["hip"]["loss"] - res["eager"]["loss"]) <= 2e-2 * abs(res["eager"]["loss"])
    # the timing is recorded, not gated: a wall-clock ratio on a shared box is no test oracle
    assert speedup > 0.0
