"""GPU: the one-launch resident-grid loss (csrc/clip_fused.hip) through its C ABI, against the numpy oracle.

The kernel's workgroups hand tile statistics and gradient tiles to each other inside one launch; the cases below cover
the arithmetic (edges off the 64 grid, gathered / partial / duplicated pairings, several weighted pairs, negative and
large scales, unnormalised rows), the hand-off protocol's re-arming (many calls on one workspace, bit-identical results)
and the host plumbing (workspace pool, forward without backward).
"""

import numpy as np
import pytest
import torch

from oracle import clip_oracle as co

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "needs a MI355X"
    return torch.device("cuda", 0)


def _bf16_round(x: np.ndarray) -> np.ndarray:
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).bfloat16().float().numpy()


def _run_fused(mats, pairs, scale, dtype, want_grad=True, upstream=1.0):
    """mats: {name: np [rows, d]}; pairs: [(ma, mb, idx_a|None, idx_b|None, n, weight)] -> loss, grads per modality, dscale."""
    from mmlearn_amd import kernels as K

    dev = _dev()
    tdt = torch.bfloat16 if dtype == "bfloat16" else torch.float32
    t = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(dev, tdt).contiguous() for k, v in mats.items()}
    d = next(iter(t.values())).shape[1]
    plan = K.clip_fused_plan(dev, [p[4] for p in pairs], d, tdt)
    assert plan is not None, "fused path refused the shape"
    s = torch.tensor([scale], dtype=torch.float32, device=dev)
    fp = []
    for ma, mb, ia, ib, n, w in pairs:
        ia_t = None if ia is None else torch.from_numpy(np.asarray(ia, np.int32)).to(dev)
        ib_t = None if ib is None else torch.from_numpy(np.asarray(ib, np.int32)).to(dev)
        fp.append((t[ma], t[mb], ia_t, ib_t, n, w))
    loss, run = K.clip_fused_forward(plan, fp, d, s, want_grad)
    out = {"loss": float(loss)}
    if not want_grad:
        del run
        return out
    grads = {k: torch.zeros(v.shape, dtype=torch.float32, device=dev) for k, v in t.items()}   # accumulate everywhere: simplest
    up = torch.tensor([upstream], dtype=torch.float32, device=dev)
    ds = torch.zeros(1, dtype=torch.float32, device=dev)
    K.clip_fused_backward(run, [(grads[p[0]], grads[p[1]], True, True) for p in pairs], s, up, ds)
    out["grads"] = {k: g.cpu().numpy() for k, g in grads.items()}
    out["dscale"] = float(ds)
    out["plan"] = plan
    return out


def _oracle(mats, ids, pairs_spec, scale, dtype):
    embs = {k: (_bf16_round(v) if True else v) for k, v in mats.items()}   # the kernel multiplies bf16-rounded rows
    return co.contrastive_loss(embs, ids, scale, pairs_spec)


def _ids(idx):
    idx = np.asarray(idx, np.int64)
    return np.stack([np.zeros_like(idx), idx], 1)


def _check(got, ref, tol, tag=""):
    assert abs(got["loss"] - ref["loss"]) <= tol * max(1.0, abs(ref["loss"])), (tag, got["loss"], ref["loss"])
    for m, g in ref["grads"].items():
        err = np.abs(got["grads"][m] - g).max()
        assert err <= tol * max(np.abs(g).max(), 1e-6), (tag, m, err, np.abs(g).max())
    assert abs(got["dscale"] - ref["dscale"]) <= tol * max(1.0, abs(ref["dscale"])), (tag, got["dscale"], ref["dscale"])


def _unit(g, n, d):
    x = g.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


@pytest.mark.parametrize("n,d,dtype,scale", [(1024, 512, "float32", 1 / 0.07), (1024, 512, "bfloat16", 1 / 0.07), (200, 96, "float32", 1 / 0.07),
                                             (37, 24, "float32", 10.0), (64, 64, "bfloat16", 30.0), (65, 72, "float32", 1 / 0.07),
                                             (1000, 768, "bfloat16", 100.0), (513, 128, "float32", -5.0), (1, 8, "float32", 3.0)])
def test_identity_pairing_vs_oracle(n, d, dtype, scale):
    g = np.random.default_rng(n * 7 + d)
    a = _unit(g, n, d)
    b = 0.6 * a + 0.8 * _unit(g, n, d)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    got = _run_fused({"rgb": a, "text": b}, [("rgb", "text", None, None, n, 1.0)], scale, dtype, upstream=0.75)
    ref = _oracle({"rgb": a, "text": b}, {"rgb": _ids(range(n)), "text": _ids(range(n))}, [(("rgb", "text"), 1.0)], scale, dtype)
    ref = {"loss": ref["loss"], "grads": {k: 0.75 * v for k, v in ref["grads"].items()}, "dscale": 0.75 * ref["dscale"]}
    _check(got, ref, 1e-2, (n, d, dtype))


def test_unnormalised_rows_and_tiny_scale():
    g = np.random.default_rng(5)
    a = 3.0 * g.standard_normal((300, 40)).astype(np.float32)
    b = 0.5 * g.standard_normal((300, 40)).astype(np.float32)
    got = _run_fused({"rgb": a, "text": b}, [("rgb", "text", None, None, 300, 0.35)], 2.0, "float32")
    ref = _oracle({"rgb": a, "text": b}, {"rgb": _ids(range(300)), "text": _ids(range(300))}, [(("rgb", "text"), 0.35)], 2.0, "float32")
    _check(got, ref, 1e-2)


@pytest.mark.parametrize("kind", ["shuffled", "partial", "duplicates"])
def test_gathered_pairings_vs_oracle(kind):
    g = np.random.default_rng(11)
    n, d = 333, 200
    a, b = _unit(g, n, d), _unit(g, n, d)
    ia = np.arange(n)
    if kind == "shuffled":
        ib = g.permutation(n)
    elif kind == "partial":
        ib = np.arange(n)
        ib[::3] += 10000
    else:
        ia = np.sort(g.integers(0, 150, n))
        ib = np.sort(g.integers(0, 150, n))
    ids = {"rgb": _ids(ia), "text": _ids(ib)}
    ma, mb = co.find_matching_indices(ids["rgb"], ids["text"])
    if len(ma) > 1024:   # heavy duplication: keep it inside the one-launch limit
        pytest.skip("too many matches for the fused path")
    got = _run_fused({"rgb": a, "text": b}, [("rgb", "text", ma, mb, len(ma), 1.0)], 1 / 0.07, "float32")
    ref = _oracle({"rgb": a, "text": b}, ids, [(("rgb", "text"), 1.0)], 1 / 0.07, "float32")
    _check(got, ref, 1e-2, kind)


def test_three_weighted_pairs_in_one_launch():
    g = np.random.default_rng(3)
    n, d = 256, 512
    r, t, au = _unit(g, n, d), _unit(g, n, d), _unit(g, 192, d)
    ids = {"rgb": _ids(range(n)), "text": _ids(g.permutation(n)), "audio": _ids(np.arange(192) * 2 % 300)}
    spec = [(("rgb", "text"), 1.0), (("rgb", "audio"), 0.5), (("text", "audio"), 0.25)]
    mats = {"rgb": r, "text": t, "audio": au}
    pairs = []
    for (ma, mb), w in spec:
        ia, ib = co.find_matching_indices(ids[ma], ids[mb])
        pairs.append((ma, mb, ia, ib, len(ia), w))
    got = _run_fused(mats, pairs, 1 / 0.07, "bfloat16")
    ref = _oracle(mats, ids, spec, 1 / 0.07, "bfloat16")
    _check(got, ref, 1e-2)


def test_many_calls_on_one_workspace_are_bit_identical_and_forward_only_works():
    """The counters of the hand-off protocol are zero again after every launch: 30 calls on one pooled workspace give the same
    bits; a forward-only call (evaluation) in between neither disturbs it nor leaks the workspace."""
    from mmlearn_amd import kernels as K

    g = np.random.default_rng(9)
    n, d = 1024, 512
    a, b = _unit(g, n, d), _unit(g, n, d)
    first = None
    for it in range(30):
        got = _run_fused({"rgb": a, "text": b}, [("rgb", "text", None, None, n, 1.0)], 1 / 0.07, "float32")
        if it % 7 == 3:
            ev = _run_fused({"rgb": a, "text": b}, [("rgb", "text", None, None, n, 1.0)], 1 / 0.07, "float32", want_grad=False)
            assert ev["loss"] == got["loss"]
        if first is None:
            first = got
        else:
            assert got["loss"] == first["loss"] and got["dscale"] == first["dscale"]
            assert np.array_equal(got["grads"]["rgb"], first["grads"]["rgb"]) and np.array_equal(got["grads"]["text"], first["grads"]["text"])
    assert len(first["plan"].pool) <= 2   # every call was served from the pool (a run gives its workspace back when it is dropped)


def test_shapes_outside_the_one_launch_path_are_refused():
    from mmlearn_amd import kernels as K

    dev = _dev()
    assert K.clip_fused_plan(dev, [1025], 512, torch.float32) is None
    assert K.clip_fused_plan(dev, [512], 510, torch.float32) is None         # rows are not whole 16-byte pieces
    assert K.clip_fused_plan(dev, [512], 516, torch.bfloat16) is None
    assert K.clip_fused_plan(dev, [512], 512, torch.float16) is None
    assert K.clip_fused_plan(dev, [256] * 5, 512, torch.float32) is None
    big = K.clip_fused_plan(dev, [1024, 1024, 1024], 512, torch.float32)     # 768 tiles: not co-resident on 256 CUs x 2
    assert big is None
    ok = K.clip_fused_plan(dev, [1024, 1024], 512, torch.float32)
    assert ok is not None and ok.grid == 512 <= ok.capacity


def test_l2_normalize_twin_feeds_the_one_launch_loss_with_the_same_bits(monkeypatch):
    """Under bf16 autocast ``ops.l2_normalize`` leaves, next to its f32 result, the rows rounded to bf16 (``_mmk_bf16_nograd``); the
    one-launch loss reads that copy instead of rounding the f32 rows itself: same loss, same gradients, bit for bit."""
    import mmlearn_amd.losses as L
    from mmlearn_amd import kernels as K
    from mmlearn_amd import ops

    dev = _dev()
    g = torch.Generator().manual_seed(21)
    raw = {m: torch.randn(1024, 512, generator=g).to(dev).bfloat16() for m in ("rgb", "text")}
    ids = torch.stack([torch.zeros(1024, dtype=torch.long), torch.arange(1024)], 1).to(dev)
    seen = []
    real = K.clip_fused_forward
    monkeypatch.setattr(K, "clip_fused_forward", lambda plan, pairs, *a, **k: (seen.append(pairs[0][0].dtype), real(plan, pairs, *a, **k))[1])
    out = []
    for use_twin in (True, False):
        leaves = {m: t.clone().requires_grad_(True) for m, t in raw.items()}
        s = torch.tensor(1 / 0.07, device=dev, requires_grad=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            emb = {m: ops.l2_normalize(t) for m, t in leaves.items()}
            for m, y in emb.items():
                assert y.dtype == torch.float32 and torch.equal(y._mmk_bf16_nograd[0], y.detach().bfloat16())
                assert not hasattr(y, "_mmk_bf16")   # that name is the DIFFERENTIABLE twin fused.linear picks up (add_layer_norm)
                if not use_twin:
                    del y._mmk_bf16_nograd
            loss = L.ContrastiveLoss()({f"{m}_embedding": y for m, y in emb.items()}, {m: ids for m in emb}, s,
                                       [L.LossPairSpec(("rgb", "text"), 1.0)])
        loss.backward()
        out.append((loss.detach().clone(), s.grad.clone(), {m: t.grad.clone() for m, t in leaves.items()}))
    assert seen == [torch.bfloat16, torch.float32]
    (l0, d0, g0), (l1, d1, g1) = out
    assert torch.equal(l0, l1) and torch.equal(d0, d1)
    for m in g0:
        assert torch.equal(g0[m], g1[m])


def test_l2_normalize_twin_is_invisible_to_linear_and_ignored_after_an_in_place_edit():
    """the loss-only copy carries no gradient, so (a) ``fused.linear`` behind ``l2_normalize`` must read the f32 rows
    (the encoder gets the gradient ``F.linear`` gives it), and (b) after an in-place edit of the normalised rows the loss must not
    read the stale copy."""
    import torch.nn.functional as F

    import mmlearn_amd.losses as L
    from mmlearn_amd import fused, ops

    dev = _dev()
    torch.manual_seed(4)
    lin = torch.nn.Linear(512, 256).to(dev)
    x0 = torch.randn(8192, 512, device=dev)
    grads = []
    for custom in (True, False):
        lin.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = ops.l2_normalize(x)
            assert fused._wgrad_linear_ok(lin.weight, y)
            out = fused.linear(y, lin.weight, lin.bias) if custom else F.linear(y, lin.weight, lin.bias)
        out.float().square().sum().backward()
        assert x.grad is not None and x.grad.abs().max().item() > 0
        grads.append(x.grad.clone())
    assert (grads[0] - grads[1]).abs().max().item() <= 2e-2 * grads[1].abs().max().item()

    ids = torch.stack([torch.zeros(512, dtype=torch.long), torch.arange(512)], 1).to(dev)
    s = torch.tensor(1 / 0.07, device=dev)
    g = torch.Generator().manual_seed(5)
    raw = {m: torch.randn(512, 64, generator=g).to(dev) for m in ("rgb", "text")}
    losses = []
    for edit_in_place in (True, False):
        with torch.autocast("cuda", dtype=torch.bfloat16), torch.no_grad():
            emb = {m: ops.l2_normalize(t) for m, t in raw.items()}
            if edit_in_place:
                emb["rgb"].mul_(-1.0)          # the copy attached to emb["rgb"] is now stale
            else:
                emb["rgb"] = -emb["rgb"]       # a new tensor: no copy attached
            losses.append(L.ContrastiveLoss()({f"{m}_embedding": y for m, y in emb.items()}, {m: ids for m in emb}, s,
                                              [L.LossPairSpec(("rgb", "text"), 1.0)]).item())
    assert abs(losses[0] - losses[1]) <= 1e-3 * abs(losses[1]), losses


def test_two_streams_launching_the_resident_grid_at_once_both_get_the_right_loss():
    """The one-launch kernel needs all its workgroups on the chip together; two of them started from different streams at the
    same moment could starve each other into the spin bound (NaN).  Launches are chained per device: both streams get the
    single-stream result, every time."""
    import mmlearn_amd.losses as L

    dev = _dev()
    g = torch.Generator().manual_seed(8)
    embs = [{m: torch.nn.functional.normalize(torch.randn(1024, 512, generator=g), dim=-1).to(dev).bfloat16() for m in ("rgb", "text")}
            for _ in range(2)]
    ids = torch.stack([torch.zeros(1024, dtype=torch.long), torch.arange(1024)], 1).to(dev)
    s = torch.tensor(1 / 0.07, device=dev)
    loss_fn = L.ContrastiveLoss()
    pairs = [L.LossPairSpec(("rgb", "text"), 1.0)]

    def one(e):
        return loss_fn({f"{m}_embedding": y for m, y in e.items()}, {m: ids for m in e}, s, pairs, fully_paired=True)

    ref = [one(e).item() for e in embs]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for s_ in streams:
        s_.wait_stream(torch.cuda.current_stream())
    for _ in range(20):
        out = []
        for k in (0, 1, 0, 1):
            with torch.cuda.stream(streams[k]):
                out.append((k, one(embs[k])))
        torch.cuda.synchronize()
        for k, v in out:
            assert torch.isfinite(v).item() and abs(v.item() - ref[k]) <= 1e-6 * abs(ref[k]), (k, v.item(), ref[k])


def test_workspaces_are_per_stream_and_a_second_backward_repeats_the_first():
    import mmlearn_amd.losses as L
    from mmlearn_amd import kernels as K

    dev = _dev()
    g = torch.Generator().manual_seed(2)
    a = torch.nn.functional.normalize(torch.randn(320, 64, generator=g), dim=-1).to(dev)
    b = torch.nn.functional.normalize(torch.randn(320, 64, generator=g), dim=-1).to(dev)
    s = torch.tensor([10.0], device=dev)
    plan = K.clip_fused_plan(dev, [320], 64, torch.float32)
    plan.pool.clear()
    side = torch.cuda.Stream()
    loss0, run0 = K.clip_fused_forward(plan, [(a, b, None, None, 320, 1.0)], 64, s, False)
    run0.release()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        loss1, run1 = K.clip_fused_forward(plan, [(a, b, None, None, 320, 1.0)], 64, s, False)
        assert len(plan.pool) == 1           # the main stream's workspace stayed in the pool: another stream got its own
        run1.release()
    torch.cuda.current_stream().wait_stream(side)
    assert len(plan.pool) == 2 and {h for h, _ in plan.pool} == {torch.cuda.current_stream().cuda_stream, side.cuda_stream}
    assert float(loss0) == float(loss1)

    ea, eb = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ids = torch.stack([torch.zeros(320, dtype=torch.long), torch.arange(320)], 1).to(dev)
    sc = torch.tensor(10.0, device=dev, requires_grad=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = L.ContrastiveLoss()({"rgb_embedding": ea, "text_embedding": eb}, {"rgb": ids, "text": ids}, sc, [L.LossPairSpec(("rgb", "text"))])
    # retain_graph: the kernel's raw gradient sums stay in the run's workspace until the graph lets go of it, so a second
    # backward repeats the finalize launch and autograd accumulates it
    loss.backward(retain_graph=True)
    g1 = (ea.grad.clone(), eb.grad.clone(), sc.grad.clone())
    loss.backward()
    assert torch.equal(ea.grad, 2 * g1[0]) and torch.equal(eb.grad, 2 * g1[1]) and torch.equal(sc.grad, 2 * g1[2])


def test_three_pairs_of_1024_rows_run_as_two_launches(monkeypatch):
    """768 tiles do not fit 512 resident slots: ``ContrastiveLoss`` groups the pairs (2 + 1) and runs the one-launch kernel twice;
    loss, every embedding gradient (two writers per modality, across the two launches) and d loss / d scale against the oracle."""
    import mmlearn_amd.losses as L
    from mmlearn_amd import kernels as K

    dev = _dev()
    g = np.random.default_rng(17)
    n, d = 1024, 256
    mats = {m: _unit(g, n, d) for m in ("rgb", "text", "audio")}
    perm = g.permutation(n)
    ids = {"rgb": _ids(range(n)), "text": _ids(perm), "audio": _ids(range(n))}
    spec = [(("rgb", "text"), 1.0), (("rgb", "audio"), 0.5), (("text", "audio"), 0.25)]
    calls = []
    real = K.clip_fused_forward
    monkeypatch.setattr(K, "clip_fused_forward", lambda plan, pairs, *a, **k: (calls.append(len(pairs)), real(plan, pairs, *a, **k))[1])
    emb = {m: torch.from_numpy(v).to(dev).requires_grad_(True) for m, v in mats.items()}
    s = torch.tensor(1 / 0.07, device=dev, requires_grad=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = L.ContrastiveLoss()({f"{m}_embedding": t for m, t in emb.items()}, {m: torch.from_numpy(v).to(dev) for m, v in ids.items()}, s,
                                   [L.LossPairSpec(m, w) for m, w in spec])
    loss.backward()
    assert calls == [2, 1]
    ref = _oracle(mats, ids, spec, 1 / 0.07, "float32")
    got = {"loss": float(loss.detach()), "grads": {m: t.grad.cpu().numpy() for m, t in emb.items()}, "dscale": float(s.grad)}
    _check(got, ref, 1e-2)


def test_one_workspace_serves_every_row_count_of_a_tile_count():
    """Plans (and their workspace pools) are keyed by tile counts, not rows: a pairing whose matched-row count changes from batch
    to batch reuses one workspace, and whatever an earlier, larger problem left in the padding must not leak into a smaller one."""
    from mmlearn_amd import kernels as K

    g = np.random.default_rng(23)
    d = 96
    plans = set()
    for n in (1000, 961, 1024, 970, 1000):          # all 16 x 16 tiles
        a = _unit(g, n, d)
        b = 0.5 * a + _unit(g, n, d)
        got = _run_fused({"rgb": a, "text": b}, [("rgb", "text", None, None, n, 1.0)], 1 / 0.07, "float32")
        ref = _oracle({"rgb": a, "text": b}, {"rgb": _ids(range(n)), "text": _ids(range(n))}, [(("rgb", "text"), 1.0)], 1 / 0.07, "float32")
        _check(got, ref, 1e-2, n)
        plans.add(id(got["plan"]))
        got = None
    assert len(plans) == 1
    plan = K.clip_fused_plan(_dev(), [1000], d, torch.float32)
    assert plan.allocs <= 2, plan.allocs
