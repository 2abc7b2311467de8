"""Pin the numpy oracle against vectors produced by running the reference itself.

CPU-only.  The golden files were written by ``tests/golden/gen_golden.py`` in the
build container (where /root/reference is importable); nothing here touches the
reference at run time.
"""

import numpy as np
import pytest
import torch

from conftest import Golden, parse_flags, parse_pairs
from oracle import clip_oracle as co
from oracle import ijepa_oracle as io

CLIP = Golden("g1_g2_clip")
MATCH = Golden("g4_match")
DIST = Golden("g3_clip_dist")
MASKS = Golden("g7_masks")


def _mods(case):
    return sorted(k[3:] for k in case if k.startswith("in_"))


@pytest.mark.parametrize("name", CLIP.names())
def test_clip_oracle_vs_reference(name):
    c = CLIP[name]
    dtype = str(c["dtype"])
    mods = _mods(c)
    embs = {m: c[f"in_{m}"] for m in mods}
    ids = {m: c[f"ids_{m}"] for m in mods}
    flags = parse_flags(c["flags"])
    res = co.contrastive_loss(embs, ids, float(c["scale"]), parse_pairs(c["pairs"]), l2norm=flags.get("l2_normalize", False))
    # the reference computed in `dtype`; the oracle in float64 on the same (already rounded) inputs
    tol = {"float32": 2e-5, "float16": 2e-3, "bfloat16": 3e-2}[dtype]
    assert bool(c["out_loss_requires_grad"]) == res["has_graph"]
    assert abs(res["loss"] - float(c["out_loss"])) <= tol * max(1.0, abs(res["loss"]))
    if res["has_graph"]:
        for m in mods:
            ref = c[f"out_grad_{m}"]
            assert np.abs(res["grads"][m] - ref).max() <= tol * max(1e-3, np.abs(ref).max()) + 1e-7
        assert abs(res["dscale"] - float(c["out_grad_scale"])) <= tol * max(1.0, abs(res["dscale"]))


@pytest.mark.parametrize("name", MATCH.names())
def test_match_oracle_vs_reference(name):
    c = MATCH[name]
    ia, ib = co.find_matching_indices(c["a"].reshape(-1, 2), c["b"].reshape(-1, 2))
    np.testing.assert_array_equal(ia, c["ia"])
    np.testing.assert_array_equal(ib, c["ib"])


def test_match_reference_known_answers():
    # tests/datasets/test_example.py:149-179 of the reference (known answers restated)
    ia, ib = co.find_matching_indices(np.array([(0, 0), (0, 1), (1, 0), (1, 1)]), np.array([(1, 0), (1, 1), (2, 0), (2, 1), (2, 2)]))
    assert ia.tolist() == [2, 3] and ib.tolist() == [0, 1]
    with pytest.raises(TypeError):
        co.find_matching_indices([(0, 0)], np.zeros((1, 2), np.int64))
    with pytest.raises(ValueError):
        co.find_matching_indices(np.zeros((3,), np.int64), np.zeros((1, 2), np.int64))


@pytest.mark.parametrize("name", DIST.names())
def test_clip_dist_oracle_vs_reference(name):
    c = DIST[name]
    W = int(c["world"])
    embs, ids = [], []
    for r in range(W):
        mods = sorted(k[len(f"r{r}_in_"):] for k in c if k.startswith(f"r{r}_in_"))
        embs.append({m: c[f"r{r}_in_{m}"] for m in mods})
        ids.append({m: c[f"r{r}_ids_{m}"] for m in mods})
    pairs = parse_pairs(c["pairs"]) if "pairs" in c else [(("rgb", "text"), 1.0)]   # n3*: three weighted pairs
    res = co.contrastive_loss_dist(embs, ids, float(c["scale"]), pairs, bool(c["local_loss"]), bool(c["gather_with_grad"]))
    for r in range(W):
        assert abs(res[r]["loss"] - float(c[f"r{r}_out_loss"])) <= 2e-5 * max(1.0, abs(res[r]["loss"])), (name, r)
        assert bool(c[f"r{r}_out_loss_requires_grad"]) == res[r]["has_graph"]
        for m in embs[r]:
            ref = c[f"r{r}_out_grad_{m}"]
            assert np.abs(res[r]["grads"][m] - ref).max() <= 2e-5 * max(1e-3, np.abs(ref).max()), (name, r, m)
        assert abs(res[r]["dscale"] - float(c[f"r{r}_out_grad_scale"])) <= 2e-5 * max(1.0, abs(res[r]["dscale"])), (name, r)


def test_ijepa_ops_oracle_vs_reference():
    c = Golden("g6_ijepa")["ops"]
    h = c["h"].astype(np.float64)
    pm, em = list(c["pred_masks"]), list(c["enc_masks"])
    np.testing.assert_array_equal(io.apply_masks(c["h"], pm), c["apply_pred"])
    np.testing.assert_array_equal(io.apply_masks(c["h"], em), c["apply_enc"])
    np.testing.assert_array_equal(io.apply_masks(c["h"], [c["per_sample_mask"]]), c["apply_per_sample"])
    np.testing.assert_allclose(io.ijepa_target(h, pm, len(em)), c["target"], atol=2e-5)
    l, dz = io.smooth_l1(c["z_pred"].astype(np.float64), c["target"].astype(np.float64))
    assert abs(l - float(c["loss_smooth_l1"])) < 1e-6
    np.testing.assert_allclose(dz, c["dz_smooth_l1"], atol=1e-9)
    l, dz = io.mse(c["z_pred"].astype(np.float64), c["target"].astype(np.float64))
    assert abs(l - float(c["loss_mse"])) < 1e-6
    np.testing.assert_allclose(dz, c["dz_mse"], atol=1e-9)
    np.testing.assert_array_equal(io.repeat_interleave_batch(c["rib_in"], 4, 2), c["rib_b4_r2"])
    np.testing.assert_array_equal(io.repeat_interleave_batch(c["rib_in"], 3, 3), c["rib_b3_r3"])
    # index form == mask form
    idx = io.masks_to_indices(pm[0])
    np.testing.assert_array_equal(np.take_along_axis(c["h"], idx[:, :, None].astype(np.int64), 1), c["apply_pred"][: h.shape[0]])


def test_predictor_assembly_oracle_vs_reference():
    c = Golden("g6_ijepa")["predictor"]
    seq = io.predictor_assemble(c["x_embed"], c["w::predictor_pos_embed"], c["w::mask_token"], list(c["enc_masks"]), list(c["pred_masks"]))
    np.testing.assert_allclose(seq, c["assembled"], atol=1e-6)


@pytest.mark.parametrize("name", MASKS.names())
def test_mask_generator_oracle_vs_reference(name):
    c = MASKS[name]
    if name.startswith("seed"):
        seed, b = name[4:].split("_b")
        torch.manual_seed(int(seed))
        m = io.ijepa_masks(batch_size=int(b))
    else:
        torch.manual_seed(4)
        m = io.ijepa_masks(batch_size=2, input_size=(96, 128), patch_size=8, npred=2, nenc=2)
    np.testing.assert_array_equal(np.stack(m["encoder_masks"]), c["enc"])
    np.testing.assert_array_equal(np.stack(m["predictor_masks"]), c["pred"])
    assert torch.randint(0, 2**31, (1,)).item() == int(c["rng_after"])


def test_ema_oracle_vs_reference():
    c = Golden("g8_ema")["copy_quirk"]
    init = {k[len("init::"):]: v for k, v in c.items() if k.startswith("init::")}
    ema = io.EmaOracle(init, 0.9, 1.0, 4)
    decays, nups = [ema.decay], [ema.num_updates]
    for step in range(6):
        student = {k.split("::", 2)[2]: v for k, v in c.items() if k.startswith(f"step{step}::student::")}
        ema.step(student)
        decays.append(ema.decay)
        nups.append(ema.num_updates)
        for k, v in ema.state.items():
            np.testing.assert_array_equal(v, c[f"step{step}::teacher::{k}"], err_msg=f"{step} {k}")
    np.testing.assert_allclose(decays, c["decays"], rtol=0, atol=1e-15)
    np.testing.assert_array_equal(nups, c["num_updates"])
    np.testing.assert_allclose([io.annealed_rate(0.996, 1.0, s, 1000) for s in (0, 1, 10, 500, 999, 1000)], c["annealed"], atol=1e-15)


ALIGN = Golden("g9_align")


@pytest.mark.parametrize("name", ALIGN.names())
def test_alignment_oracle_vs_reference(name):
    c = ALIGN[name]
    order = c["order"].tolist()
    if name.startswith("w2_"):
        embs = [{m: c[f"r{r}_in_{m}"] for m in order} for r in range(2)]
        ids = [{m: c[f"r{r}_ids_{m}"] for m in order} for r in range(2)]
        res = co.contrastive_loss_dist(embs, ids, float(c["scale"]), [(("rgb", "text"), 1.0)], bool(c["local_loss"]),
                                       bool(c["gather_with_grad"]), modality_alignment=True)
        for r in range(2):
            assert abs(res[r]["loss"] - float(c[f"r{r}_out_loss"])) <= 2e-5 * abs(res[r]["loss"])
            for m in order:
                ref = c[f"r{r}_out_grad_{m}"]
                assert np.abs(res[r]["grads"][m] - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1e-3), (name, r, m)
            assert abs(res[r]["dscale"] - float(c[f"r{r}_out_grad_scale"])) <= 2e-5 * max(1.0, abs(res[r]["dscale"]))
        return
    embs = {m: c[f"in_{m}"] for m in order}
    ids = {m: c[f"ids_{m}"] for m in order}
    for prefix, pairs in (("out", parse_pairs(c["pairs"])), ("only", [])):
        res = co.contrastive_loss(embs, ids, float(c["scale"]), pairs, modality_alignment=True)
        assert abs(res["loss"] - float(c[f"{prefix}_loss"])) <= 2e-5 * abs(res["loss"]), (name, prefix)
        for m in order:
            ref = c[f"{prefix}_grad_{m}"]
            assert np.abs(res["grads"][m] - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1e-3), (name, prefix, m)
        assert abs(res["dscale"] - float(c[f"{prefix}_grad_scale"])) <= 2e-5 * max(1.0, abs(res["dscale"]))


def test_attention_dropout_mask_statistics():
    """The counter-based keep-mask restated in oracle/attention_oracle.py: rate, independence along i / j / (b, h)."""
    from oracle import attention_oracle as AO

    m = AO.keep_mask(987654321, 4, 8, 197, 0.1).astype(np.float64)
    assert abs(m.mean() - (1 - AO.drop_threshold(0.1) / 65536)) < 2e-3
    d = m - m.mean()
    var = d.var()
    for a, b in ((d[..., :-1], d[..., 1:]), (d[..., :-1, :], d[..., 1:, :]), (d[:, :-1], d[:, 1:]), (d[:-1], d[1:])):
        assert abs((a * b).mean() / var) < 5e-3
    assert AO.keep_mask(1, 1, 1, 8, 0.0).all()
    assert (AO.keep_mask(5, 1, 2, 64, 0.5) != AO.keep_mask(6, 1, 2, 64, 0.5)).mean() > 0.4


def test_retrieval_recall_oracle_matches_reference_golden():
    """G10: oracle/metrics_oracle.py against RetrievalRecallAtK.compute() of the reference (all cases, k, aggregations)."""
    import os

    from oracle import metrics_oracle as mo

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g10_recall.npz"))
    cases = sorted({k.split("/")[0] for k in g.files})
    assert len(cases) == 4
    for c in cases:
        x, y, idx, bs = g[c + "/in_x"], g[c + "/in_y"], g[c + "/in_indexes"], int(g[c + "/in_batch"])
        n = len(x)
        X, Y, I = mo.concat_batches([x[s:s + bs] for s in range(0, n, bs)], [y[s:s + bs] for s in range(0, n, bs)],
                                    [idx[s:s + bs] for s in range(0, n, bs)])
        outs = [f for f in g.files if f.startswith(c + "/out_")]
        assert outs
        for f in outs:
            _, k, agg = f.split("/")[1].split("_")[1:]
            assert abs(mo.recall_at_k(X, Y, I, int(k[1:]), agg) - float(g[f])) < 1e-6, f
    # tie rule: a duplicate of the positive with a LOWER index outranks it, one with a higher index does not
    x = np.eye(4, dtype=np.float32)
    y = np.stack([x[1], x[0], x[0], x[3]]).astype(np.float32)     # query 0's positive is y row 2, duplicated at row 1
    assert mo.ranks(x[:1], y, np.array([2])).tolist() == [1] and mo.ranks(x[:1], y, np.array([1])).tolist() == [0]
