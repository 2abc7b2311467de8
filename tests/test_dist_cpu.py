"""World-size > 1 tests on CPU (gloo): the row-sharded orchestration of ``mmlearn_amd.losses`` (packed
all-gather, ownership, label offsets, LSE/loss all-reduce, per-flag gradient recipes) against the
per-rank outputs of the REFERENCE itself under torch.distributed (golden g3_clip_dist).

The kernels are replaced by the CPU test double ``tests/fake_kernels.py`` (monkeypatched into
``mmlearn_amd.losses.K`` inside the worker processes only); the HIP kernels are covered by -m gpu tests.
"""

import os
import sys
import traceback

import numpy as np
import pytest
import torch
import torch.distributed as dist

import mp_util
from conftest import Golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIST = Golden("g3_clip_dist")


def _worker(rank, world, port, case_names, q, done):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
            if p not in sys.path:
                sys.path.insert(0, p)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import fake_kernels
        import mmlearn_amd.losses as L
        from conftest import Golden as G

        L.K = fake_kernels  # test seam: CPU double of the kernel layer
        gold = G("g3_clip_dist")
        results = {}
        for name in case_names:
            c = gold[name]
            mods = sorted(k[len(f"r{rank}_in_"):] for k in c if k.startswith(f"r{rank}_in_"))
            embs = {f"{m}_embedding": torch.tensor(c[f"r{rank}_in_{m}"]).requires_grad_(True) for m in mods}
            ids = {m: torch.tensor(c[f"r{rank}_ids_{m}"]) for m in mods}
            s = torch.tensor(float(c["scale"]), requires_grad=True)
            from conftest import parse_pairs
            specs = [L.LossPairSpec(m, w) for m, w in (parse_pairs(c["pairs"]) if "pairs" in c else [(("rgb", "text"), 1.0)])]
            from mmlearn_amd.wire import pairing_summary
            hint, _ = pairing_summary(ids)   # what the collator would put into batch["fully_paired"] on this rank
            said = [None] * world
            dist.all_gather_object(said, (hint, tuple(mods)))
            flags = [h and m == said[0][1] for h, m in said]   # ... and every rank holds the same modalities
            for static in ((False, True, "paired") if "uneven" not in name and "missing" not in name else (False, "paired")):
                for t in embs.values():
                    t.grad = None
                s.grad = None
                fn = L.ContrastiveLoss(local_loss=bool(c["local_loss"]), gather_with_grad=bool(c["gather_with_grad"]),
                                       static_shapes=static is True)
                # this variant also takes the "backward keeps G on chip" packing branch: no transposed operand may be asked for
                fake_kernels.ON_CHIP_BACKWARD = static == "paired"
                if static == "paired":   # wire-format hint: identity pairing only when EVERY rank's batch is paired
                    before, before_t = fake_kernels.CALLS["match_ids"], fake_kernels.CALLS["transposes"]
                    loss = fn(embs, ids, s, specs, fully_paired=hint)
                    assert fake_kernels.CALLS["transposes"] == before_t or "alignment" in name, "transposed operands packed for the one-kernel backward"
                    if all(flags):
                        assert fake_kernels.CALLS["match_ids"] == before, "paired batch must not run the matcher"
                    else:
                        assert fake_kernels.CALLS["match_ids"] > before or len(mods) < 2
                    results.setdefault("_paired_ranks", {})[name] = all(flags)
                    rec = {"loss": float(loss.detach()), "requires_grad": loss.requires_grad}
                    if loss.requires_grad:
                        loss.backward()
                    rec["grads"] = {m: (embs[f"{m}_embedding"].grad.numpy().copy() if embs[f"{m}_embedding"].grad is not None
                                        else np.zeros_like(c[f"r{rank}_in_{m}"])) for m in mods}
                    rec["dscale"] = float(s.grad) if s.grad is not None else 0.0
                    results[(name, static)] = rec
                    continue
                if static:  # the task starts the gathers right after each encoder; the loss must pick them up
                    for m in mods:
                        fn.prefetch_gather(m, embs[f"{m}_embedding"], ids[m])
                    assert set(fn._pending) == set(mods)
                loss = fn(embs, ids, s, specs)
                assert not fn._pending
                rec = {"loss": float(loss.detach()), "requires_grad": loss.requires_grad}
                if loss.requires_grad:
                    loss.backward()
                rec["grads"] = {m: (embs[f"{m}_embedding"].grad.numpy().copy() if embs[f"{m}_embedding"].grad is not None
                                    else np.zeros_like(c[f"r{rank}_in_{m}"])) for m in mods}
                rec["dscale"] = float(s.grad) if s.grad is not None else 0.0
                results[(name, static)] = rec
        item = (rank, results, None)
    except Exception:  # surface the traceback in the parent
        item = (rank, None, traceback.format_exc())
    try:
        mp_util.send(q, done, item)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _collect(items):
    out = {}
    for rank, res, err in items:
        assert err is None, f"rank {rank} failed:\n{err}"
        out[rank] = res
    return out


def _run(world, case_names):
    return _collect(mp_util.run(_worker, world, lambda r, port: (r, world, port, case_names), timeout=240))


def _check(world, case_names, out):
    for name in case_names:
        c = DIST[name]
        local = bool(c["local_loss"])
        for rank in range(world):
            for static in (False, True, "paired"):
                if (name, static) not in out[rank]:
                    continue
                got = out[rank][(name, static)]
                tag = (name, rank, static)
                ref_loss = float(c[f"r{rank}_out_loss"])
                ref_has_graph = bool(c[f"r{rank}_out_loss_requires_grad"])
                assert abs(got["loss"] - ref_loss) <= 2e-5 * max(1.0, abs(ref_loss)), (tag, got["loss"], ref_loss)
                assert got["requires_grad"] or not ref_has_graph, tag
                for m, g in got["grads"].items():
                    ref = c[f"r{rank}_out_grad_{m}"] if ref_has_graph else np.zeros_like(g)
                    assert np.abs(g - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1e-3), (tag, m, np.abs(g - ref).max())
                ref_ds = float(c[f"r{rank}_out_grad_scale"]) if ref_has_graph else 0.0
                assert abs(got["dscale"] - ref_ds) <= 2e-5 * max(1.0, abs(ref_ds)), (tag, got["dscale"], ref_ds, local)


@pytest.mark.timeout(300)
def test_world2_all_flag_cells_uneven_and_missing_modality():
    names = [n for n in DIST.names() if n.startswith("w2_")]
    assert len(names) == 9
    out = _run(2, names)
    _check(2, names, out)
    paired = out[0]["_paired_ranks"]
    assert any(paired.values()) and not all(paired.values()), paired   # both the fast path and its refusal were exercised


def _align_worker(rank, world, port, q, done):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
            if p not in sys.path:
                sys.path.insert(0, p)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import fake_kernels
        import mmlearn_amd.losses as L
        from conftest import Golden as G

        L.K = fake_kernels
        gold = G("g9_align")
        results = {}
        for name in [n for n in gold.names() if n.startswith("w2_")]:
            c = gold[name]
            embs = {f"{m}_embedding": torch.tensor(c[f"r{rank}_in_{m}"]).requires_grad_(True) for m in ("rgb", "text")}
            ids = {m: torch.tensor(c[f"r{rank}_ids_{m}"]) for m in ("rgb", "text")}
            s = torch.tensor(float(c["scale"]), requires_grad=True)
            fn = L.ContrastiveLoss(local_loss=bool(c["local_loss"]), gather_with_grad=bool(c["gather_with_grad"]),
                                   modality_alignment=True)
            loss = fn(embs, ids, s, [L.LossPairSpec(("rgb", "text"))])
            loss.backward()
            results[name] = {"loss": float(loss.detach()), "dscale": float(s.grad),
                             "grads": {m: embs[f"{m}_embedding"].grad.numpy().copy() for m in ("rgb", "text")}}
        item = (rank, results, None)
    except Exception:
        item = (rank, None, traceback.format_exc())
    try:
        mp_util.send(q, done, item)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world2_modality_alignment_all_flag_cells():
    gold = Golden("g9_align")
    out = _collect(mp_util.run(_align_worker, 2, lambda r, port: (r, 2, port), timeout=240))
    for name in [n for n in gold.names() if n.startswith("w2_")]:
        c = gold[name]
        for rank in range(2):
            got = out[rank][name]
            assert abs(got["loss"] - float(c[f"r{rank}_out_loss"])) <= 2e-5 * abs(got["loss"]), (name, rank)
            for m in ("rgb", "text"):
                ref = c[f"r{rank}_out_grad_{m}"]
                assert np.abs(got["grads"][m] - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1e-3), (name, rank, m)
            assert abs(got["dscale"] - float(c[f"r{rank}_out_grad_scale"])) <= 2e-5 * max(1.0, abs(got["dscale"])), (name, rank)


@pytest.mark.timeout(300)
def test_world4_all_flag_cells():
    names = [n for n in DIST.names() if n.startswith("w4_")]
    assert len(names) == 4
    out = _run(4, names)
    _check(4, names, out)


def _static_worker(rank, world, port, q, done):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
            if p not in sys.path:
                sys.path.insert(0, p)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import fake_kernels
        import mmlearn_amd.losses as L

        L.K = fake_kernels
        res = {}
        torch.manual_seed(rank)

        def batch(b, mods=("rgb", "text")):
            embs = {f"{m}_embedding": torch.randn(b, 8, dtype=torch.float64).requires_grad_(True) for m in mods}
            ids = {m: torch.stack([torch.zeros(b, dtype=torch.long), torch.arange(rank * 100, rank * 100 + b)], 1) for m in mods}
            return embs, ids

        s = torch.tensor(3.0, requires_grad=True)
        pairs = [L.LossPairSpec(("rgb", "text"))]
        fn = L.ContrastiveLoss(static_shapes=True, local_loss=True, gather_with_grad=True)
        # (1) equal shapes: validated once, then no header collective (and, the global pairing being the identity, no
        #     per-pair row-count exchange either)
        calls = {"n": 0}
        orig = L._all_gather

        def counting(t, w):
            calls["n"] += 1
            return orig(t, w)

        L._all_gather = counting
        for step in range(3):
            embs, ids = batch(6)
            fn(embs, ids, s, pairs).backward()
            res[f"gathers_step{step}"] = calls["n"]
            calls["n"] = 0
        # (2) a shorter last batch on EVERY rank: re-validated, passes
        embs, ids = batch(4)
        fn(embs, ids, s, pairs).backward()
        res["short_ok"] = True
        # (3) one rank with a different batch size / without a modality: a clear error on every rank instead of a hang
        for tag, (b, mods) in {"rows": (5 if rank == 1 else 7, ("rgb", "text")), "missing": (7, ("rgb", "text") if rank == 0 else ("rgb",))}.items():
            embs, ids = batch(b, mods)
            try:
                fn(embs, ids, s, pairs)
                res[tag] = "no error"
            except ValueError as e:
                res[tag] = str(e)
        L._all_gather = orig
        item = (rank, res, None)
    except Exception:
        item = (rank, None, traceback.format_exc())
    try:
        mp_util.send(q, done, item)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world2_three_modalities_three_weighted_pairs():
    """the multi-pair exchange buffer (``exch_off`` with > 1 pair) and per-pair row ownership across ranks,
    against the reference's own per-rank outputs: 3 modalities / 3 weighted pairs, unequal rows per modality, partial pairing,
    audio rows matched across ranks (scattered ownership) and a rank without audio."""
    names = [n for n in DIST.names() if n.startswith("n3w2_")]
    assert len(names) == 8
    out = _run(2, names)
    _check(2, names, out)


@pytest.mark.timeout(300)
def test_world4_three_modalities_three_weighted_pairs():
    names = [n for n in DIST.names() if n.startswith("n3w4_")]
    assert len(names) == 4
    out = _run(4, names)
    _check(4, names, out)


@pytest.mark.timeout(600)
def test_world8_all_flag_cells():
    """SURVEY 8(c) G3 asks for W in {2, 4, 8}: eight gloo processes against the reference's eight per-rank outputs."""
    names = [n for n in DIST.names() if n.startswith("w8_")]
    assert len(names) == 4
    out = _run(8, names)
    _check(8, names, out)


@pytest.mark.timeout(300)
def test_static_shapes_is_a_checked_promise():
    """``static_shapes=True`` used to be unchecked -- a rank with a short batch or without a modality made the
    ranks issue mismatched collectives.  Now the first step (and any step on which the local shapes change) exchanges a
    header and raises on disagreement; steady-state steps add no collective, and with an identity global pairing the
    local-loss cell no longer exchanges its per-pair row counts."""
    out = _collect(mp_util.run(_static_worker, 2, lambda r, port: (r, 2, port), timeout=240))
    for rank in range(2):
        r = out[rank]
        # embeddings + ids gathers every step; the validation header only on the first
        assert r["gathers_step0"] == r["gathers_step1"] + 1 == r["gathers_step2"] + 1, r
        assert r["gathers_step1"] == 2, r
        assert r["short_ok"]
        assert "ranks disagree" in r["rows"] and "[7, 5]" in r["rows"], r["rows"]
        assert "ranks disagree" in r["missing"] and "-1" in r["missing"], r["missing"]
