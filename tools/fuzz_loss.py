"""Randomised sweep of the contrastive loss against the numpy oracle (run by hand on the GPU; the oracle is test infrastructure).
Random row counts off and on the tile grid, widths, dtypes, scales on both sides of the one-exponential bound, 2-3 modalities,
partial / shuffled / duplicated ids, l2_normalize; TN_MIN_ROWS=256 in the environment also sends mid-size mirrored pairs
through the transposed-read backward.    N=40 SEED=0 python tools/fuzz_loss.py"""
import os, random, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import ContrastiveLoss, LossPairSpec
from mmlearn_amd import kernels as K
from oracle import clip_oracle as co

if os.environ.get("TN_MIN_ROWS"):
    K.TN_MIN_ROWS = int(os.environ["TN_MIN_ROWS"])

dev = torch.device("cuda", 0)
rng = random.Random(int(os.environ.get("SEED", 0)))
bad = 0
for it in range(int(os.environ.get("N", 40))):
    n = rng.choice([rng.randint(1, 300), rng.randint(300, 1500), rng.choice([128, 256, 512, 1024])])
    d = rng.choice([8 * rng.randint(1, 80), 64, 128, 512, rng.randint(3, 200)])
    dt = rng.choice([torch.float32, torch.bfloat16, torch.bfloat16])
    scale = rng.choice([1.0, 1 / 0.07, 30.0, 100.0, -5.0])
    l2 = rng.random() < 0.3
    align = rng.random() < 0.2
    mods = ["rgb", "text", "audio"][: rng.choice([2, 2, 3])]
    g = torch.Generator().manual_seed(1000 + it)
    embs, ids = {}, {}
    base = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=-1)
    kind = rng.choice(["paired", "paired", "shuffled", "partial", "dups"])
    for m in mods:
        nm = n if kind != "partial" else max(1, n - rng.randint(0, n // 3))
        x = torch.nn.functional.normalize(0.5 * base[:nm] + torch.nn.functional.normalize(torch.randn(nm, d, generator=g), dim=-1), dim=-1)
        if not l2 and rng.random() < 0.2:
            x = x * (0.5 + 2.0 * torch.rand(nm, 1, generator=g))   # rows off the unit sphere
        idx = torch.arange(nm)
        if kind == "shuffled" and m != mods[0]:
            perm = torch.randperm(nm, generator=g)
            x, idx = x[perm], idx[perm]
        if kind == "dups" and nm > 3:
            idx = idx.clone()
            idx[rng.randrange(nm)] = idx[rng.randrange(nm)]
        embs[m] = x.to(dt)
        ids[m] = torch.stack([torch.zeros(nm, dtype=torch.long), idx], 1)
    pairs = [((a, b), rng.choice([1.0, 0.5, 2.0])) for i, a in enumerate(mods) for b in mods[i + 1:]]
    te = {f"{m}_embedding": embs[m].to(dev).requires_grad_(True) for m in mods}
    s = torch.tensor(scale, device=dev, requires_grad=True)
    try:
        loss = ContrastiveLoss(l2_normalize=l2, modality_alignment=align)(te, {m: ids[m].to(dev) for m in mods}, s, [LossPairSpec(p, w) for p, w in pairs])
        loss.float().backward()
    except Exception as e:
        bad += 1
        print("LOSS RAISED", dict(it=it, n=n, d=d, dt=str(dt), scale=scale, l2=l2, align=align, mods=len(mods), kind=kind,
                                  rows={m: tuple(embs[m].shape) for m in mods}), repr(e)[:300], flush=True)
        continue
    ref = co.contrastive_loss({m: embs[m].float().numpy() for m in mods}, {m: ids[m].numpy() for m in mods}, scale, pairs, l2norm=l2,
                              modality_alignment=align)
    # bf16 operands: the packed (normalised) rows are rounded to bf16 before the MFMA, and a rounding of 2^-8 of a cosine is
    # 0.4 in the logits at scale 100: the sharper the softmax, the more of it shows
    tol = (3e-2 if abs(scale) >= 30 else 1e-2) if dt == torch.bfloat16 else 1e-3
    errs = [abs(float(loss.detach().float()) - ref["loss"]) / max(1.0, abs(ref["loss"]))]
    for m in mods:
        gr = te[f"{m}_embedding"].grad.float().cpu().numpy()
        errs.append(np.abs(gr - ref["grads"][m]).max() / max(np.abs(ref["grads"][m]).max(), 1e-6))
    errs.append(abs(float(s.grad) - ref["dscale"]) / max(1.0, abs(ref["dscale"])))
    if max(errs) > tol or not np.isfinite(max(errs)):
        bad += 1
        print("LOSS MISMATCH", dict(it=it, n=n, d=d, dt=str(dt), scale=scale, l2=l2, align=align, mods=len(mods), kind=kind), [f"{e:.2e}" for e in errs], flush=True)
torch.cuda.synchronize()
print("fuzz_loss done, mismatches:", bad)
