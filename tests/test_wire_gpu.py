"""GPU: the wire-format hint and the packed keys give the same loss and gradients as matching on
[B, 2] ids, without the matcher launch or its read-back."""

import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(n=192, d=64, seed=3):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(n, d, generator=g).cuda().requires_grad_(True)
    b = torch.randn(n, d, generator=g).cuda().requires_grad_(True)
    s = torch.tensor(7.5, device="cuda", requires_grad=True)
    ids = torch.stack([torch.zeros(n, dtype=torch.long), torch.randperm(n, generator=g)], 1).cuda()
    return a, b, s, ids


def _run(fn, a, b, s, ids_a, ids_b, **kw):
    from mmlearn_amd import LossPairSpec
    for t in (a, b, s):
        t.grad = None
    loss = fn({"rgb_embedding": a, "text_embedding": b}, {"rgb": ids_a, "text": ids_b}, s, [LossPairSpec(("rgb", "text"))], **kw)
    loss.backward()
    return loss.detach().clone(), a.grad.clone(), b.grad.clone(), s.grad.clone()


def test_fully_paired_hint_skips_matcher_and_matches_loss():
    from mmlearn_amd import ContrastiveLoss, _lib
    from mmlearn_amd.wire import pack_example_ids
    a, b, s, ids = _case()
    fn = ContrastiveLoss(l2_normalize=True)
    want = _run(fn, a, b, s, ids, ids.clone())
    _lib.profile_enable(True)
    _lib.profile_read()
    got = _run(fn, a, b, s, ids, ids.clone(), fully_paired=True)
    torch.cuda.synchronize()
    prof = _lib.profile_read()
    _lib.profile_enable(False)
    assert not any(k.startswith("match") and v[0] for k, v in prof.items()), {k: v for k, v in prof.items() if k.startswith("match")}
    for w, g in zip(want, got):
        assert torch.equal(w, g)   # identical launches once the pairing is known: bit-equal
    packed = _run(fn, a, b, s, pack_example_ids(ids), pack_example_ids(ids))
    for w, g in zip(want, packed):
        assert torch.equal(w, g)


def test_hint_with_unequal_rows_is_refused_and_permuted_ids_still_match():
    from mmlearn_amd import ContrastiveLoss
    from mmlearn_amd.wire import pairing_summary
    a, b, s, ids = _case(n=96)
    fn = ContrastiveLoss()
    with pytest.raises(ValueError, match="different row counts"):
        _run(fn, a, b[:64].detach().requires_grad_(True), s, ids, ids[:64], fully_paired=True)
    perm = torch.randperm(96, generator=torch.Generator().manual_seed(1)).cuda()
    ids_b = ids[perm]
    assert pairing_summary({"rgb": ids.cpu(), "text": ids_b.cpu()})[0] is False   # the collator would not set the flag
    b2 = b.detach()[perm].requires_grad_(True)
    base = _run(fn, a, b, s, ids, ids.clone())
    moved = _run(fn, a, b2, s, ids, ids_b)   # matcher undoes the permutation
    assert torch.allclose(base[0], moved[0], rtol=1e-5, atol=1e-6)
    inv = torch.argsort(perm)   # b2[j] = b[perm[j]]  =>  grad_b[i] = grad_b2[inv[i]]
    assert torch.allclose(base[2], moved[2][inv], rtol=1e-4, atol=1e-6)


def test_prefetched_matcher_answer_is_used_once_and_equals_the_synchronous_one():
    """``prefetch_match`` (matcher on its own stream before the encoders, status parked in pinned memory) must give the
    pairing the synchronous matcher gives -- incl. a permuted, partially overlapping id set -- and must not be reused
    for other tensors or a later batch."""
    from mmlearn_amd import ContrastiveLoss, LossPairSpec, _lib
    a, b, s, ids = _case(n=160)
    ids_b = ids[torch.randperm(160, generator=torch.Generator().manual_seed(5)).cuda()].clone()
    ids_b[:7, 1] += 1000          # seven rows of b have no partner
    pairs = [LossPairSpec(("rgb", "text"))]
    fn = ContrastiveLoss(l2_normalize=True)
    want = _run(fn, a, b, s, ids, ids_b)
    _lib.profile_enable(True)
    _lib.profile_read()
    fn.prefetch_match({"rgb": ids, "text": ids_b}, pairs)
    assert len(fn._pending_match) == 1
    got = _run(fn, a, b, s, ids, ids_b)
    torch.cuda.synchronize()
    prof = _lib.profile_read()
    _lib.profile_enable(False)
    assert prof["match_ids"][0] == 1 and not fn._pending_match     # one launch: the prefetched one
    for w, g in zip(want, got):
        assert torch.equal(w, g)
    # answers for other tensors are not picked up: same values, different storage -> the synchronous matcher runs
    fn.prefetch_match({"rgb": ids, "text": ids_b}, pairs)
    other = _run(fn, a, b, s, ids.clone(), ids_b.clone())
    assert not fn._pending_match
    for w, g in zip(want, other):
        assert torch.equal(w, g)
