"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference):

    python tests/golden/gen_golden.py

Writes small ``.npz`` fixtures (inputs + the reference's outputs).  The
fixtures are data; no reference source is stored.  Fixture ids follow
SURVEY.md §8(c): G1 clip-2mod, G2 clip-Nmod, G3 clip-dist, G4 match,
G5 task-step, G6 ijepa-ops, G7 masks, G8 ema; G9 alignment loss, G10 retrieval recall@k (the
SURVEY 8(f3) widening).
"""

from __future__ import annotations

import os
import sys
import warnings

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
import tiny_models  # noqa: E402

warnings.filterwarnings("ignore")


def _np(t):
    if isinstance(t, torch.Tensor):
        t = t.detach()
        if t.dtype in (torch.bfloat16, torch.float16):
            t = t.float()
        return t.cpu().numpy().copy()
    return np.array(t)


def _ids(idx, ds=0):
    idx = torch.as_tensor(idx, dtype=torch.long)
    return torch.stack([torch.full_like(idx, ds), idx], 1)


# --------------------------------------------------------------------- G1/G2
def _run_loss(R, embs, ids, scale, pairs, **flags):
    """embs: {mod: tensor}, ids: {mod: tensor}; returns loss + grads."""
    embs_g = {k: v.clone().requires_grad_(True) for k, v in embs.items()}
    s = torch.tensor(float(scale), requires_grad=True)
    loss_fn = R.ContrastiveLoss(**flags)
    specs = [R.LossPairSpec(modalities=tuple(p[0]), weight=float(p[1])) for p in pairs]
    loss = loss_fn({f"{k}_embedding": v for k, v in embs_g.items()}, ids, s, specs)
    out = {"loss": _np(loss), "loss_requires_grad": np.array(loss.requires_grad)}
    if loss.requires_grad:
        loss.float().backward()
        for k, v in embs_g.items():
            out[f"grad_{k}"] = _np(v.grad if v.grad is not None else torch.zeros_like(v))
        out["grad_scale"] = _np(s.grad if s.grad is not None else torch.zeros(()))
    return out


def gen_clip(R):
    cases = {}
    g = torch.Generator().manual_seed(1234)

    def rn(*shape):
        return torch.randn(*shape, generator=g)

    def add(name, embs, ids, scale, pairs, dtype="float32", **flags):
        tdt = {"float32": torch.float32, "bfloat16": torch.bfloat16, "float16": torch.float16}[dtype]
        embs = {k: v.to(tdt) for k, v in embs.items()}
        out = _run_loss(R, embs, ids, scale, pairs, **flags)
        rec = {f"in_{k}": _np(v) for k, v in embs.items()}
        rec.update({f"ids_{k}": _np(v) for k, v in ids.items()})
        rec["scale"] = np.array(scale, dtype=np.float64)
        rec["pairs"] = np.array([f"{p[0][0]}|{p[0][1]}|{p[1]}" for p in pairs])
        rec["dtype"] = np.array(dtype)
        rec["flags"] = np.array([f"{k}={v}" for k, v in flags.items()])
        rec.update({f"out_{k}": v for k, v in out.items()})
        cases[name] = rec

    s0 = 1 / 0.07
    pair_rt = [(("rgb", "text"), 1.0)]
    # config-1-sized, paired
    a, b = F.normalize(rn(64, 512), dim=-1), F.normalize(rn(64, 512), dim=-1)
    add("c64x512_paired", {"rgb": a, "text": b}, {"rgb": _ids(range(64)), "text": _ids(range(64))}, s0, pair_rt)
    # small variants
    a, b = F.normalize(rn(16, 32), dim=-1), F.normalize(rn(16, 32), dim=-1)
    perm = torch.randperm(16, generator=g)
    add("s16_shuffled", {"rgb": a, "text": b}, {"rgb": _ids(range(16)), "text": _ids(perm)}, s0, pair_rt)
    ids_b = torch.arange(16)
    ids_b[::3] += 100  # unmatched texts
    add("s16_partial", {"rgb": a, "text": b}, {"rgb": _ids(range(16)), "text": _ids(ids_b)}, s0, pair_rt)
    ids_a = torch.tensor([0, 1, 2, 2, 3, 4, 5, 5, 6, 7, 8, 9, 10, 11, 12, 13])
    ids_b = torch.tensor([0, 1, 2, 3, 3, 4, 5, 6, 7, 8, 9, 9, 10, 11, 12, 40])
    add("s16_duplicates", {"rgb": a, "text": b}, {"rgb": _ids(ids_a), "text": _ids(ids_b)}, s0, pair_rt)
    add("s16_weight", {"rgb": a, "text": b}, {"rgb": _ids(range(16)), "text": _ids(range(16))}, 30.0, [(("rgb", "text"), 0.35)])
    a3, b3 = 3.0 * rn(16, 32), 0.5 * rn(16, 32)
    add("s16_l2norm_flag", {"rgb": a3, "text": b3}, {"rgb": _ids(range(16)), "text": _ids(range(16))}, s0, pair_rt, l2_normalize=True)
    add("s16_unnormalized", {"rgb": 0.3 * a3, "text": b3}, {"rgb": _ids(range(16)), "text": _ids(range(16))}, 2.0, pair_rt)
    add("s16_bf16", {"rgb": a, "text": b}, {"rgb": _ids(range(16)), "text": _ids(range(16))}, s0, pair_rt, dtype="bfloat16")
    add("s16_fp16", {"rgb": a, "text": b}, {"rgb": _ids(range(16)), "text": _ids(range(16))}, s0, pair_rt, dtype="float16")
    add("s16_scale100", {"rgb": a, "text": b}, {"rgb": _ids(range(16)), "text": _ids(range(16))}, 100.0, pair_rt)
    add("s16_nomatch", {"rgb": a, "text": b}, {"rgb": _ids(range(16)), "text": _ids(range(100, 116))}, s0, pair_rt)
    # different dataset index => no match even with equal example index
    add("s16_dataset_index", {"rgb": a, "text": b}, {"rgb": _ids(range(16), 0), "text": torch.cat([_ids(range(8), 0), _ids(range(8, 16), 1)])}, s0, pair_rt)
    # ragged: N != M
    add("s12x16_ragged", {"rgb": a[:12], "text": b}, {"rgb": _ids(range(12)), "text": _ids(range(16))}, s0, pair_rt)
    # odd sizes that do not align with any kernel tile
    a5, b5 = F.normalize(rn(37, 24), dim=-1), F.normalize(rn(37, 24), dim=-1)
    add("s37x24_odd", {"rgb": a5, "text": b5}, {"rgb": _ids(range(37)), "text": _ids(range(37))}, s0, pair_rt)
    a6, b6 = F.normalize(rn(200, 96), dim=-1), F.normalize(rn(200, 96), dim=-1)
    add("s200x96", {"rgb": a6, "text": b6}, {"rgb": _ids(range(200)), "text": _ids(range(200))}, s0, pair_rt)

    # G2: three modalities, partial overlap
    r, t, au = F.normalize(rn(12, 32), dim=-1), F.normalize(rn(12, 32), dim=-1), F.normalize(rn(8, 32), dim=-1)
    ids3 = {"rgb": _ids(range(12)), "text": _ids([0, 1, 2, 3, 4, 5, 20, 21, 22, 9, 10, 11]), "audio": _ids([2, 3, 4, 5, 6, 7, 30, 31])}
    add("n3_explicit", {"rgb": r, "text": t, "audio": au}, ids3, s0, [(("rgb", "text"), 1.0), (("rgb", "audio"), 0.5), (("text", "audio"), 0.25)])
    add("n3_default", {"rgb": r, "text": t, "audio": au}, ids3, s0, [(("rgb", "text"), 1.0), (("rgb", "audio"), 1.0), (("text", "audio"), 1.0)])
    add("n3_one_pair_missing", {"rgb": r, "text": t}, {"rgb": ids3["rgb"], "text": ids3["text"]}, s0, [(("rgb", "text"), 1.0), (("rgb", "audio"), 1.0)])
    return cases


# ------------------------------------------------------------------------ G3
def _dist_worker(rank, world, port, cfg, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    warnings.filterwarnings("ignore")
    R = ref_shim.load()
    torch.manual_seed(100 + rank)
    B = cfg["B"][rank]
    D = cfg["D"]
    mods = cfg["mods"][rank]
    if "rows" in cfg:   # N-way cases: rows per modality, explicit id columns per (rank, modality)
        embs = {m: F.normalize(torch.randn(cfg["rows"][rank][m], D), dim=-1) for m in mods}
        ids = {m: _ids(cfg["ids"][rank][m]) for m in mods}
    else:
        embs = {m: F.normalize(torch.randn(B, D), dim=-1) for m in mods}
        ids = {m: _ids(range(cfg["off"][rank], cfg["off"][rank] + B)) for m in mods}
    out = _run_loss(
        R, embs, ids, cfg["scale"], cfg.get("pairs", [((("rgb", "text")), 1.0)]),
        local_loss=cfg["local_loss"], gather_with_grad=cfg["gather_with_grad"],
    )
    rec = {f"in_{k}": _np(v) for k, v in embs.items()}
    rec.update({f"ids_{k}": _np(v) for k, v in ids.items()})
    rec.update({f"out_{k}": v for k, v in out.items()})
    q.put((rank, rec))
    dist.barrier()
    dist.destroy_process_group()


N3_PAIRS = [(("rgb", "text"), 1.0), (("rgb", "audio"), 0.5), (("text", "audio"), 0.25)]


def _nway_cfgs():
    """3 modalities / 3 weighted pairs across ranks (BASELINE configs[3] shape of the loss: the reference loops
    modality_loss_pairs over the GATHERED embeddings, contrastive.py:92-144).  Every rank: 8 rgb + 8 text rows with the
    same ids, 6 audio rows that match a subset of them (partial pairing inside the rank, unequal rows per modality)."""
    cfgs = []
    for world in (2, 4):
        rows = [{"rgb": 8, "text": 8, "audio": 6}] * world
        ids = [{"rgb": list(range(8 * r, 8 * r + 8)), "text": list(range(8 * r, 8 * r + 8)),
                "audio": [8 * r + 1, 8 * r + 2, 8 * r + 4, 8 * r + 5, 8 * r + 7, 1000 + r]} for r in range(world)]
        for ll in (False, True):
            for gwg in (False, True):
                cfgs.append(dict(name=f"n3w{world}_local{int(ll)}_gwg{int(gwg)}", world=world, B=[8] * world, D=16,
                                 mods=[["rgb", "text", "audio"]] * world, rows=rows, ids=ids, pairs=N3_PAIRS, scale=1 / 0.07,
                                 local_loss=ll, gather_with_grad=gwg))
    # audio rows matched ACROSS ranks (a global permutation of the example ids): owned rows of the audio side are
    # scattered over the matched list.  Full-batch cells only (with local_loss the reference slices arange labels by
    # local match counts, which presumes in-rank pairing).
    world = 2
    perm = [5, 12, 0, 9, 14, 3, 7, 10, 1, 15, 6, 11, 2, 13, 4, 8]
    rows = [{"rgb": 8, "text": 8, "audio": 8}] * world
    ids = [{"rgb": list(range(8 * r, 8 * r + 8)), "text": list(range(8 * r, 8 * r + 8)), "audio": perm[8 * r: 8 * r + 8]}
           for r in range(world)]
    for gwg in (False, True):
        cfgs.append(dict(name=f"n3w2_scatter_local0_gwg{int(gwg)}", world=world, B=[8] * world, D=16,
                         mods=[["rgb", "text", "audio"]] * world, rows=rows, ids=ids, pairs=N3_PAIRS, scale=1 / 0.07,
                         local_loss=False, gather_with_grad=gwg))
    # rank 1 has no audio at all (placeholder path of _gather_dicts, contrastive.py:471-480); local_loss=False only (Q3)
    rows = [{"rgb": 8, "text": 8, "audio": 6}, {"rgb": 8, "text": 8}]
    ids = [{"rgb": list(range(8)), "text": list(range(8)), "audio": [1, 2, 4, 5, 7, 1000]},
           {"rgb": list(range(8, 16)), "text": list(range(8, 16))}]
    for gwg in (False, True):
        cfgs.append(dict(name=f"n3w2_missing_audio_local0_gwg{int(gwg)}", world=2, B=[8, 8], D=16,
                         mods=[["rgb", "text", "audio"], ["rgb", "text"]], rows=rows, ids=ids, pairs=N3_PAIRS, scale=1 / 0.07,
                         local_loss=False, gather_with_grad=gwg))
    return cfgs


def gen_dist():
    cases = {}
    port = 29610
    ctx = mp.get_context("spawn")
    cfgs = []
    for world in (2, 4):
        for ll in (False, True):
            for gwg in (False, True):
                cfgs.append(dict(name=f"w{world}_local{int(ll)}_gwg{int(gwg)}", world=world, B=[8] * world, D=16,
                                 mods=[["rgb", "text"]] * world, off=[8 * r for r in range(world)], scale=1 / 0.07,
                                 local_loss=ll, gather_with_grad=gwg))
    # unequal per-rank batch (pad path)
    for ll in (False, True):
        for gwg in (False, True):
            cfgs.append(dict(name=f"w2_uneven_local{int(ll)}_gwg{int(gwg)}", world=2, B=[8, 5], D=16,
                             mods=[["rgb", "text"]] * 2, off=[0, 8], scale=1 / 0.07, local_loss=ll, gather_with_grad=gwg))
    # a rank missing a modality: only the cells where the reference neither hangs nor
    # returns a graph-less loss that would break a joint backward (Q3): local_loss=False
    for gwg in (False,):
        cfgs.append(dict(name=f"w2_missing_local0_gwg{int(gwg)}", world=2, B=[8, 8], D=16,
                         mods=[["rgb", "text"], ["rgb"]], off=[0, 8], scale=1 / 0.07, local_loss=False, gather_with_grad=gwg))
    # W = 8 (eight gloo processes, B = 8, D = 16, the four flag cells)
    for ll in (False, True):
        for gwg in (False, True):
            cfgs.append(dict(name=f"w8_local{int(ll)}_gwg{int(gwg)}", world=8, B=[8] * 8, D=16,
                             mods=[["rgb", "text"]] * 8, off=[8 * r for r in range(8)], scale=1 / 0.07,
                             local_loss=ll, gather_with_grad=gwg))
    cfgs += _nway_cfgs()
    only = os.environ.get("GEN_DIST_ONLY")   # debugging aid: regenerate a subset (do not save a partial file)
    for cfg in cfgs:
        if only and only not in cfg["name"]:
            continue
        port += 1
        q = ctx.Queue()
        procs = [ctx.Process(target=_dist_worker, args=(r, cfg["world"], port, cfg, q)) for r in range(cfg["world"])]
        for p in procs:
            p.start()
        res = dict(q.get(timeout=300) for _ in procs)
        for p in procs:
            p.join(timeout=60)
        rec = {"world": np.array(cfg["world"]), "local_loss": np.array(cfg["local_loss"]),
               "gather_with_grad": np.array(cfg["gather_with_grad"]), "scale": np.array(cfg["scale"])}
        if "pairs" in cfg:
            rec["pairs"] = np.array([f"{p[0][0]}|{p[0][1]}|{p[1]}" for p in cfg["pairs"]])
        for r, d in res.items():
            for k, v in d.items():
                rec[f"r{r}_{k}"] = v
        cases[cfg["name"]] = rec
        print("  dist", cfg["name"], [float(res[r]["out_loss"]) for r in sorted(res)], flush=True)
    return cases


# ------------------------------------------------------------------------ G4
def gen_match(R):
    cases = {}

    def add(name, a, b):
        ia, ib = R.find_matching_indices(a, b)
        cases[name] = {"a": _np(a), "b": _np(b), "ia": _np(ia), "ib": _np(ib)}

    # the reference's own known-answer cases (tests/datasets/test_example.py:149-179)
    add("ref_basic", torch.tensor([(0, 0), (0, 1), (1, 0), (1, 1)]), torch.tensor([(1, 0), (1, 1), (2, 0), (2, 1), (2, 2)]))
    add("ref_dups", torch.tensor([(0, 0), (0, 1), (1, 0), (1, 1), (0, 0)]), torch.tensor([(0, 0), (1, 1), (0, 0), (2, 2)]))
    add("ref_nomatch", torch.tensor([(0, 0), (0, 1)]), torch.tensor([(1, 0), (1, 1)]))
    g = torch.Generator().manual_seed(7)
    a = torch.stack([torch.randint(0, 3, (300,), generator=g), torch.randint(0, 40, (300,), generator=g)], 1)
    b = torch.stack([torch.randint(0, 3, (257,), generator=g), torch.randint(0, 40, (257,), generator=g)], 1)
    add("rand_dups_300x257", a, b)
    add("identity_1000", _ids(range(1000)), _ids(range(1000)))
    add("perm_513", _ids(torch.randperm(513, generator=g)), _ids(torch.randperm(513, generator=g)))
    add("empty_a", torch.zeros(0, 2, dtype=torch.long), _ids(range(4)))
    big = torch.tensor([(2**40 + 3, -5), (7, 2**33), (2**40 + 3, -5)])
    add("wide_values", big, torch.tensor([(7, 2**33), (2**40 + 3, -5), (7, 2**33 + 1)]))
    return cases


# ------------------------------------------------------------------------ G5
def gen_task(R):
    cases = {}
    for name, init_log_scale in (("default", None), ("clamp_hi", 6.0), ("clamp_lo", -1.0)):
        torch.manual_seed(5)
        B, D = 16, 32
        enc = {"rgb": tiny_models.FlatMLPEncoder("rgb", 3 * 8 * 8, 48, D), "text": tiny_models.TokenMLPEncoder("text", 50, 24, D)}
        task = R.ContrastivePretraining(encoders=enc, loss=R.ContrastiveLoss(), compute_validation_loss=False, compute_test_loss=False)
        if init_log_scale is not None:
            with torch.no_grad():
                task.log_logit_scale.fill_(init_log_scale)
        batch = {"rgb": torch.rand(B, 3, 8, 8), "text": torch.randint(0, 50, (B, 77)),
                 "example_ids": {"rgb": _ids(range(B)), "text": _ids(range(B))}}
        state = {k: _np(v) for k, v in task.state_dict().items()}
        loss = task.training_step(batch, 0)
        loss.backward()
        rec = {f"w::{k}": v for k, v in state.items()}
        rec.update({"rgb": _np(batch["rgb"]), "text": _np(batch["text"]), "loss": _np(loss),
                    "log_logit_scale_after": _np(task.log_logit_scale),
                    "logged_logit_scale": _np(task.logged["train/logit_scale"]),
                    "logged_loss": _np(task.logged["train/loss"])})
        for k, p in task.named_parameters():
            rec[f"g::{k}"] = _np(p.grad if p.grad is not None else torch.zeros_like(p))
        cases[name] = rec
    return cases


# --------------------------------------------------------------------- G6/G7
def gen_ijepa(R):
    cases = {}
    torch.manual_seed(11)
    B, N, D = 4, 196, 64
    h = torch.randn(B, N, D) * 1.7 + 0.3
    torch.manual_seed(3)
    mi = R.IJEPAMaskGenerator()(batch_size=B)
    enc_masks, pred_masks = mi["encoder_masks"], mi["predictor_masks"]
    hm = R.apply_masks(h, pred_masks)
    rec = {"h": _np(h), "enc_masks": _np(torch.stack(enc_masks)), "pred_masks": _np(torch.stack(pred_masks)),
           "apply_pred": _np(hm), "apply_enc": _np(R.apply_masks(h, enc_masks))}
    hn = F.layer_norm(h, h.size()[-1:])
    tgt = R.repeat_interleave_batch(R.apply_masks(hn, pred_masks), B, repeat=len(enc_masks))
    rec["target"] = _np(tgt)
    z = (tgt + 0.8 * torch.randn_like(tgt)).requires_grad_(True)
    loss = F.smooth_l1_loss(z, tgt)
    loss.backward()
    rec.update({"z_pred": _np(z), "loss_smooth_l1": _np(loss), "dz_smooth_l1": _np(z.grad)})
    z2 = z.detach().clone().requires_grad_(True)
    l2 = F.mse_loss(z2, tgt)
    l2.backward()
    rec.update({"loss_mse": _np(l2), "dz_mse": _np(z2.grad)})
    x = torch.arange(24.0).view(12, 2)
    rec["rib_in"] = _np(x)
    rec["rib_b4_r2"] = _np(R.repeat_interleave_batch(x, 4, 2))
    rec["rib_b3_r3"] = _np(R.repeat_interleave_batch(x, 3, 3))
    # per-sample (B, N) masks with equal keep counts
    g = torch.Generator().manual_seed(9)
    pm = torch.zeros(B, N, dtype=torch.int32)
    for bi in range(B):
        pm[bi, torch.randperm(N, generator=g)[:20]] = 1
    rec["per_sample_mask"] = _np(pm)
    rec["apply_per_sample"] = _np(R.apply_masks(h, [pm]))
    cases["ops"] = rec

    # predictor assembly with a tiny predictor (fixed weights)
    torch.manual_seed(21)
    P = R.VisionTransformerPredictor(num_patches=196, embed_dim=D, predictor_embed_dim=32, depth=0, num_heads=2)
    with torch.no_grad():
        P.mask_token.normal_(0, 0.5)
    zc = R.apply_masks(torch.randn(B, N, D), enc_masks).requires_grad_(True)
    out = P(zc, enc_masks, pred_masks)
    out.square().mean().backward()
    prec = {f"w::{k}": _np(v) for k, v in P.state_dict().items()}
    prec.update({"z_ctx": _np(zc), "enc_masks": rec["enc_masks"], "pred_masks": rec["pred_masks"], "out": _np(out),
                 "dz_ctx": _np(zc.grad), "d_mask_token": _np(P.mask_token.grad)})
    # the assembled sequence itself (pre-blocks), rebuilt with the reference helpers
    with torch.no_grad():
        xe = P.predictor_embed(zc.detach())
        xe = xe + R.apply_masks(P.predictor_pos_embed.repeat(B, 1, 1), enc_masks)
        pe = R.repeat_interleave_batch(R.apply_masks(P.predictor_pos_embed.repeat(B, 1, 1), pred_masks), B, repeat=len(enc_masks))
        seq = torch.cat([xe.repeat(len(pred_masks), 1, 1), P.mask_token.repeat(pe.size(0), pe.size(1), 1) + pe], dim=1)
    prec["assembled"] = _np(seq)
    prec["x_embed"] = _np(P.predictor_embed(zc.detach()))
    cases["predictor"] = prec

    # full IJEPA training_step with tiny ViT
    torch.manual_seed(31)
    enc = tiny_models.TinyPatchEncoder(embed_dim=32, mask_fn=R.apply_masks)
    pred = R.VisionTransformerPredictor(num_patches=196, embed_dim=32, predictor_embed_dim=16, depth=0, num_heads=2)
    with torch.no_grad():
        pred.mask_token.normal_(0, 0.5)
    task = R.IJEPA(encoder=enc, predictor=pred)
    task.configure_model()
    imgs = torch.rand(2, 3, 224, 224)
    torch.manual_seed(77)
    loss = task.training_step({"rgb": imgs}, 0)
    loss.backward()
    srec = {f"w::{k}": _np(v) for k, v in task.state_dict().items()}
    srec.update({"images_seed": np.array(31), "images": _np(imgs).astype(np.float16), "loss": _np(loss),
                 "mask_seed": np.array(77), "ema_decay_logged": np.array(task.logged["train/ema_decay"])})
    gsum = {k: float(p.grad.abs().sum()) for k, p in task.named_parameters() if p.grad is not None}
    srec["grad_abs_sums"] = np.array([f"{k}={v:.8e}" for k, v in gsum.items()])
    cases["step"] = srec
    return cases


def gen_masks(R):
    cases = {}
    for seed in (0, 1, 2, 3, 12345):
        for B in (1, 3):
            torch.manual_seed(seed)
            mi = R.IJEPAMaskGenerator()(batch_size=B)
            after = torch.randint(0, 2**31, (1,)).item()  # pins how far the global RNG advanced
            cases[f"seed{seed}_b{B}"] = {"enc": _np(torch.stack(mi["encoder_masks"])), "pred": _np(torch.stack(mi["predictor_masks"])),
                                         "rng_after": np.array(after)}
    torch.manual_seed(4)
    mi = R.IJEPAMaskGenerator(input_size=(96, 128), patch_size=8, npred=2, nenc=2)(batch_size=2)
    cases["custom_96x128_p8"] = {"enc": _np(torch.stack(mi["encoder_masks"])), "pred": _np(torch.stack(mi["predictor_masks"])),
                                 "rng_after": np.array(torch.randint(0, 2**31, (1,)).item())}
    return cases


# ------------------------------------------------------------------------ G8
def gen_ema(R):
    cases = {}
    torch.manual_seed(8)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 3))
    ema = R.ExponentialMovingAverage(net, 0.9, 1.0, 4)
    ema.configure_model("cpu")
    rec = {}
    for k, v in net.state_dict().items():
        rec[f"init::{k}"] = _np(v)
    decays, nups = [ema.decay], [ema.num_updates]
    for step in range(6):
        with torch.no_grad():
            for i, p in enumerate(net.parameters()):
                p.add_(0.1 * (step + 1) * (i + 1))
            net[1].running_mean.add_(0.5)
            net[1].num_batches_tracked.add_(1)
        ema.step(net)
        decays.append(ema.decay)
        nups.append(ema.num_updates)
        for k, v in ema.model.state_dict().items():
            rec[f"step{step}::teacher::{k}"] = _np(v)
        for k, v in net.state_dict().items():
            rec[f"step{step}::student::{k}"] = _np(v)
    rec["decays"] = np.array(decays)
    rec["num_updates"] = np.array(nups)
    rec["annealed"] = np.array([R.ExponentialMovingAverage.get_annealed_rate(0.996, 1.0, s, 1000) for s in (0, 1, 10, 500, 999, 1000)])
    cases["copy_quirk"] = rec
    return cases


# ------------------------------------------------------------------------ G9
def _align_worker(rank, world, port, cfg, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    warnings.filterwarnings("ignore")
    R = ref_shim.load()
    torch.manual_seed(300 + rank)
    embs = {m: F.normalize(torch.randn(n, cfg["D"]), dim=-1) for m, n in cfg["sizes"].items()}
    ids = {m: _ids(range(rank * 50, rank * 50 + n)) for m, n in cfg["sizes"].items()}
    out = _run_loss(R, embs, ids, cfg["scale"], cfg["pairs"], modality_alignment=True,
                    local_loss=cfg["local_loss"], gather_with_grad=cfg["gather_with_grad"])
    rec = {f"in_{k}": _np(v) for k, v in embs.items()}
    rec.update({f"ids_{k}": _np(v) for k, v in ids.items()})
    rec.update({f"out_{k}": v for k, v in out.items()})
    q.put((rank, rec))
    dist.barrier()
    dist.destroy_process_group()


def gen_align(R):
    """modality_alignment=True (contrastive.py:344-413) incl. the non-cumulative offset quirk (Q2)."""
    cases = {}
    g = torch.Generator().manual_seed(99)
    for name, sizes, pairs, scale in (
        ("m2_6x6", {"rgb": 6, "text": 6}, [(("rgb", "text"), 1.0)], 1 / 0.07),
        ("m3_5x7x4", {"rgb": 5, "text": 7, "audio": 4}, [(("rgb", "text"), 1.0), (("rgb", "audio"), 0.5)], 10.0),
        ("m3_equal_8", {"rgb": 8, "text": 8, "audio": 8}, [(("rgb", "text"), 1.0), (("text", "audio"), 1.0)], 5.0),
        ("m2_40x33_d24", {"rgb": 40, "text": 33}, [(("rgb", "text"), 1.0)], 3.0),
    ):
        d = 24 if "d24" in name else 16
        embs = {m: F.normalize(torch.randn(n, d, generator=g), dim=-1) for m, n in sizes.items()}
        ids = {m: _ids(range(n)) for m, n in sizes.items()}
        out = _run_loss(R, embs, ids, scale, pairs, modality_alignment=True)
        rec = {f"in_{k}": _np(v) for k, v in embs.items()}
        rec.update({f"ids_{k}": _np(v) for k, v in ids.items()})
        rec["scale"] = np.array(scale)
        rec["pairs"] = np.array([f"{p[0][0]}|{p[0][1]}|{p[1]}" for p in pairs])
        rec["order"] = np.array(list(sizes))
        rec.update({f"out_{k}": v for k, v in out.items()})
        # the alignment term alone (no pairs -> only the alignment loss is appended)
        only = _run_loss(R, embs, ids, scale, [], modality_alignment=True)
        rec.update({f"only_{k}": v for k, v in only.items()})
        cases[name] = rec
    ctx = mp.get_context("spawn")
    port = 29660
    for ll, gwg in ((False, False), (False, True), (True, False), (True, True)):
        port += 1
        cfg = dict(D=16, sizes={"rgb": 6, "text": 6}, scale=8.0, pairs=[(("rgb", "text"), 1.0)], local_loss=ll, gather_with_grad=gwg)
        q = ctx.Queue()
        procs = [ctx.Process(target=_align_worker, args=(r, 2, port, cfg, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = dict(q.get(timeout=180) for _ in procs)
        for p in procs:
            p.join(timeout=60)
        rec = {"world": np.array(2), "local_loss": np.array(ll), "gather_with_grad": np.array(gwg), "scale": np.array(8.0),
               "order": np.array(["rgb", "text"])}
        for r, dd in res.items():
            for k, v in dd.items():
                rec[f"r{r}_{k}"] = v
        cases[f"w2_local{int(ll)}_gwg{int(gwg)}"] = rec
        print("  align dist", ll, gwg, [float(res[r]["out_loss"]) for r in sorted(res)])
    return cases


def gen_recall(R):
    """G10: RetrievalRecallAtK (modules/metrics/retrieval_recall.py) -- update() in batches, compute() on CPU."""
    import mmlearn.modules.metrics.retrieval_recall as rr

    cases = {}
    gen = torch.Generator().manual_seed(1234)
    for name, (n, d, bs, ks, mode) in {
        "paired_64": (64, 32, 16, (1, 5, 10), "paired"),
        "shuffled_96": (96, 48, 32, (1, 3, 10), "shuffled"),
        "noisy_128": (128, 64, 64, (1, 5, 20), "noisy"),
        "single_batch_40": (40, 16, 40, (1, 2, 7), "noisy"),
    }.items():
        base = torch.randn(n, d, generator=gen)
        if mode == "paired":
            x, y = base + 0.3 * torch.randn(n, d, generator=gen), base.clone()
        elif mode == "shuffled":
            x, y = base + 0.5 * torch.randn(n, d, generator=gen), base.clone()
        else:
            x, y = base + 1.5 * torch.randn(n, d, generator=gen), base + 0.2 * torch.randn(n, d, generator=gen)
        x, y = 3.0 * x, 0.5 * y  # unnormalised inputs: compute() normalises
        rec = {"in_x": _np(x), "in_y": _np(y), "in_batch": np.array(bs)}
        # within every batch the positives are a permutation of the batch's y rows (the metric offsets them by the count so far)
        idx_batches = []
        for s0 in range(0, n, bs):
            b = min(bs, n - s0)
            idx_batches.append(torch.randperm(b, generator=gen) if mode == "shuffled" else torch.arange(b))
        rec["in_indexes"] = _np(torch.cat(idx_batches))
        for k in ks:
            for agg in ("mean", "min"):
                m = rr.RetrievalRecallAtK(top_k=k, reduction="none", aggregation=agg)
                for bi, s0 in enumerate(range(0, n, bs)):
                    b = min(bs, n - s0)
                    yb = y[s0:s0 + b]
                    if mode == "shuffled":   # row i of the batch matches y row idx[i] of the same batch
                        m.update(x[s0:s0 + b], yb, idx_batches[bi])
                    else:
                        m.update(x[s0:s0 + b], yb, idx_batches[bi])
                rec[f"out_recall_k{k}_{agg}"] = _np(m.compute())
        cases[name] = rec
        print("  recall", name, {k: float(v) for k, v in rec.items() if k.startswith("out_")})
    return cases


# ------------------------------------------------------------------------ G11
def gen_wire(R):
    """Example / CombinedDataset / DefaultDataCollator outputs of the reference for tests/golden/wire_scenario.py."""
    from mmlearn.datasets.core import CombinedDataset, DefaultDataCollator, Example

    import wire_scenario
    return {"scenario": wire_scenario.run(Example, CombinedDataset, DefaultDataCollator)}


def _save(name, cases):
    flat = {}
    for c, rec in cases.items():
        for k, v in rec.items():
            flat[f"{c}/{k}"] = v
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **flat)
    print(f"wrote {path}: {len(cases)} cases, {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    R = ref_shim.load()
    which = sys.argv[1:] or ["clip", "match", "task", "ijepa", "masks", "ema", "dist", "align", "recall", "wire"]
    if "clip" in which:
        _save("g1_g2_clip", gen_clip(R))
    if "match" in which:
        _save("g4_match", gen_match(R))
    if "task" in which:
        _save("g5_task", gen_task(R))
    if "ijepa" in which:
        _save("g6_ijepa", gen_ijepa(R))
    if "masks" in which:
        _save("g7_masks", gen_masks(R))
    if "ema" in which:
        _save("g8_ema", gen_ema(R))
    if "dist" in which:
        _save("g3_clip_dist", gen_dist())
    if "align" in which:
        _save("g9_align", gen_align(R))
    if "recall" in which:
        _save("g10_recall", gen_recall(R))
    if "wire" in which:
        _save("g11_wire", gen_wire(R))


if __name__ == "__main__":
    main()
