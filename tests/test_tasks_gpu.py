"""GPU parity of whole task steps against the reference's own outputs (golden g5_task / g6_ijepa "step"):
same tiny encoders (tests/golden/tiny_models.py), same weights, same inputs and RNG seeds."""

from functools import partial

import numpy as np
import pytest
import torch

from conftest import Golden

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda", 0)


def _ids(n, dev):
    return torch.stack([torch.zeros(n, dtype=torch.long), torch.arange(n)], 1).to(dev)


@pytest.mark.parametrize("name", ["default", "clamp_hi", "clamp_lo"])
def test_contrastive_training_step_vs_reference(name):
    import tiny_models
    from mmlearn_amd import ContrastiveLoss
    from mmlearn_amd.tasks import ContrastivePretraining

    c = Golden("g5_task")[name]
    dev = _dev()
    B, D = 16, 32
    enc = {"rgb": tiny_models.FlatMLPEncoder("rgb", 3 * 8 * 8, 48, D), "text": tiny_models.TokenMLPEncoder("text", 50, 24, D)}
    task = ContrastivePretraining(encoders=enc, loss=ContrastiveLoss(), compute_validation_loss=False, compute_test_loss=False)
    task.load_state_dict({k[3:]: torch.tensor(v) for k, v in c.items() if k.startswith("w::")})
    task.to(dev)
    batch = {"rgb": torch.tensor(c["rgb"], device=dev), "text": torch.tensor(c["text"], device=dev),
             "example_ids": {"rgb": _ids(B, dev), "text": _ids(B, dev)}}
    loss = task.training_step(batch, 0)
    loss.backward()
    assert abs(loss.item() - float(c["loss"])) <= 1e-3 * max(1.0, abs(float(c["loss"])))
    # clamp happened in place, before the loss read exp() (Q6)
    assert abs(task.log_logit_scale.item() - float(c["log_logit_scale_after"])) < 1e-6
    assert abs(float(task.logged["train/logit_scale"]) - float(c["logged_logit_scale"])) <= 1e-4 * float(c["logged_logit_scale"])
    assert abs(float(task.logged["train/loss"]) - float(c["logged_loss"])) <= 1e-3 * max(1.0, abs(float(c["logged_loss"])))
    for k, p in task.named_parameters():
        ref = c[f"g::{k}"]
        got = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(ref)
        assert np.abs(got - ref).max() <= 1e-3 * max(np.abs(ref).max(), 1e-6) + 1e-7, k


def test_contrastive_task_surface():
    import tiny_models
    from mmlearn_amd import ContrastiveLoss, LearnableLogitScaling, Modalities
    from mmlearn_amd.tasks import ContrastivePretraining, LossPairSpec, ModuleKeySpec

    dev = _dev()
    D = 16
    mk = lambda: tiny_models.FlatMLPEncoder("x", 12, 8, D)  # noqa: E731

    class Enc(torch.nn.Module):
        def __init__(self, key):
            super().__init__()
            self.key, self.lin = key, torch.nn.Linear(12, D)

        def forward(self, inputs):
            return (self.lin(inputs[self.key]),)

    shared = {"proj": torch.nn.Linear(D, D), "scale": LearnableLogitScaling()}
    task = ContrastivePretraining(
        encoders={"rgb": Enc("rgb"), "text": Enc("text"), "audio": Enc("audio")},
        heads={"shared": shared},
        modality_module_mapping={m: ModuleKeySpec(encoder_key=m, head_key="shared") for m in ("rgb", "text", "audio")},
        loss=ContrastiveLoss(), optimizer=partial(torch.optim.AdamW, lr=1e-3),
        modality_loss_pairs=[LossPairSpec(("rgb", "text")), LossPairSpec(("rgb", "audio"), 0.5), LossPairSpec(("text", "audio"), 0.25)],
        compute_validation_loss=True, compute_test_loss=False).to(dev)
    # dict heads become Sequentials over the SAME module instances (Q5)
    assert task.heads["rgb"][0] is task.heads["text"][0] is task.heads["audio"][0]
    B = 24
    batch = {m: torch.randn(B, 12, device=dev) for m in ("rgb", "text", "audio")}
    batch["example_ids"] = {m: _ids(B, dev) for m in ("rgb", "text", "audio")}
    out = task(batch)
    assert set(out) == {"rgb_embedding", "text_embedding", "audio_embedding"}
    for v in out.values():
        np.testing.assert_allclose(v.norm(dim=-1).detach().cpu().numpy(), 1.0, atol=1e-5)
    opt = task.configure_optimizers()
    names = {g["name"]: len(g["params"]) for g in opt.param_groups}
    assert names["no_weight_decay_params"] > 0 and names["weight_decay_params"] > 0
    loss = task.training_step(batch, 0)
    loss.backward()
    opt.step()
    assert torch.isfinite(loss)
    # logit-scaling head is cancelled by the normalisation: ~zero gradient (Q5)
    assert task.heads["rgb"][1].log_logit_scale.grad.abs().item() < 1e-4
    vloss = task.validation_step(batch, 0)
    assert torch.isfinite(vloss) and "val/loss" in task.logged
    assert task.test_step(batch, 0) is None
    # dimension mismatch -> ValueError like the reference
    bad = ContrastivePretraining(encoders={"rgb": Enc("rgb"), "text": torch.nn.Sequential()}, loss=ContrastiveLoss(),
                                 compute_validation_loss=False, compute_test_loss=False)
    bad.encoders["text"] = type("E", (torch.nn.Module,), {"forward": lambda self, i: (torch.zeros(B, D + 1, device=dev),)})()
    with pytest.raises(ValueError, match="same dimension"):
        bad.to(dev)(batch)
    with pytest.raises(ValueError, match="unsupported modality"):
        ContrastivePretraining(encoders={"rgb": mk(), "smell": mk()}, loss=ContrastiveLoss())
    with pytest.raises(ValueError, match="Loss function must be provided"):
        ContrastivePretraining(encoders={"rgb": mk(), "text": mk()})
    assert Modalities.get_modality("rgb").embedding == "rgb_embedding"


def test_validation_step_under_inference_mode_and_bf16_autocast():
    """Lightning's validate / test / predict loops run under `torch.inference_mode()` (inference tensors track no version
    counter): `encode(normalize=True)` and the loss must run there under bf16 autocast with a twin-eligible width (a multiple of
    8), and give the value of the same step under `no_grad`."""
    import tiny_models
    from mmlearn_amd import ContrastiveLoss
    from mmlearn_amd.tasks import ContrastivePretraining

    dev = _dev()
    B, D = 64, 32
    torch.manual_seed(11)
    enc = {"rgb": tiny_models.FlatMLPEncoder("rgb", 3 * 8 * 8, 48, D), "text": tiny_models.TokenMLPEncoder("text", 50, 24, D)}
    task = ContrastivePretraining(encoders=enc, loss=ContrastiveLoss(), compute_validation_loss=True, compute_test_loss=True).to(dev).eval()
    batch = {"rgb": torch.randn(B, 3, 8, 8, device=dev), "text": torch.randint(0, 50, (B, 6), device=dev),
             "example_ids": {"rgb": _ids(B, dev), "text": _ids(B, dev)}}
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        want = float(task.validation_step(batch, 0))
    with torch.inference_mode(), torch.autocast("cuda", dtype=torch.bfloat16):
        e = task.encode(batch, task_modality("rgb"), normalize=True)
        assert e.is_inference() and e.dtype == torch.float32 and not hasattr(e, "_mmk_bf16_nograd")
        got = float(task.validation_step(batch, 0))
        assert np.isfinite(float(task.test_step(batch, 0)))
    assert abs(got - want) <= 1e-2 * max(1.0, abs(want)), (got, want)   # twin vs in-kernel rounding of the same rows: same bits expected
    with torch.inference_mode():   # f32, no autocast
        assert np.isfinite(float(task.validation_step(batch, 0)))


def task_modality(name):
    from mmlearn_amd import Modalities

    return Modalities.get_modality(name)


def test_ijepa_training_step_vs_reference():
    import tiny_models
    from mmlearn_amd import ops
    from mmlearn_amd.tasks import IJEPA

    c = Golden("g6_ijepa")["step"]
    dev = _dev()
    enc = tiny_models.TinyPatchEncoder(embed_dim=32, mask_fn=ops.apply_masks)
    pred = tiny_models.SimplePredictor(196, 32, 16)
    task = IJEPA(encoder=enc, predictor=pred)
    sd = {k[3:]: torch.tensor(v) for k, v in c.items() if k.startswith("w::")}
    task.load_state_dict(sd)
    task.target_encoder.model.load_state_dict({k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")})
    task.to(dev)
    task.configure_model()
    imgs = torch.tensor(c["images"].astype(np.float32), device=dev)
    torch.manual_seed(int(c["mask_seed"]))
    loss = task.training_step({"rgb": imgs}, 0)
    loss.backward()
    # images were stored as fp16 -> compare at 2e-3
    assert abs(loss.item() - float(c["loss"])) <= 2e-3 * max(1.0, float(c["loss"])), (loss.item(), float(c["loss"]))
    assert float(task.logged["train/ema_decay"]) == float(c["ema_decay_logged"])
    ref = dict(s.split("=") for s in c["grad_abs_sums"].tolist())
    for k, p in task.named_parameters():
        if p.grad is None:
            continue
        r = float(ref[k])
        assert abs(p.grad.abs().sum().item() - r) <= 2e-2 * max(r, 1e-6), (k, p.grad.abs().sum().item(), r)
    # EMA hook + checkpoint hooks
    before = task.target_encoder.num_updates
    task.on_before_zero_grad(None)
    assert task.target_encoder.num_updates == before + 1
    for (k, t), (_, s) in zip(task.target_encoder.model.state_dict().items(), task.encoder.state_dict().items()):
        assert torch.equal(t, s), k   # copy quirk (Q1)
    ckpt = {}
    task.on_save_checkpoint(ckpt)
    assert set(ckpt["ema_params"]) == {"decay", "num_updates"}
    task.on_load_checkpoint(dict(ckpt))
    # user loss_fn path: HIP target + arbitrary callable
    task2 = IJEPA(encoder=enc, predictor=pred, loss_fn=lambda p, t: (p - t).abs().mean())
    task2.to(dev)
    task2.configure_model()
    l2 = task2.training_step({"rgb": imgs}, 0)
    assert torch.isfinite(l2) and l2.requires_grad


def test_contrastive_with_ijepa_auxiliary():
    import tiny_models
    from mmlearn_amd import ContrastiveLoss, ops
    from mmlearn_amd.tasks import IJEPA, AuxiliaryTaskSpec, ContrastivePretraining

    dev = _dev()
    torch.manual_seed(0)

    rgb = tiny_models.TinyPatchEncoder(embed_dim=32, mask_fn=ops.apply_masks)

    class RgbForContrastive(torch.nn.Module):
        """contrastive branch pools the tokens; the I-JEPA branches (mask given, or teacher) see tokens"""

        def __init__(self, enc):
            super().__init__()
            self.enc = enc
            self.patch_embed, self.embed_dim, self.num_heads = enc.patch_embed, enc.embed_dim, enc.num_heads

        def forward(self, inputs):
            x = self.enc(inputs)[0]
            return (x if "rgb_mask" in inputs or getattr(self, "tokens", False) else x.mean(1), None)

    wrapped = RgbForContrastive(rgb)
    aux = partial(IJEPA, predictor=tiny_models.SimplePredictor(196, 32, 16), loss_fn=None)
    task = ContrastivePretraining(
        encoders={"rgb": wrapped, "text": tiny_models.TokenMLPEncoder("text", 50, 24, 32)},
        loss=ContrastiveLoss(), auxiliary_tasks={"ijepa": AuxiliaryTaskSpec(modality="rgb", task=aux, loss_weight=0.5)},
        log_auxiliary_tasks_loss=True, compute_validation_loss=False, compute_test_loss=False).to(dev)
    task.configure_model()
    task.auxiliary_tasks["ijepa"].target_encoder.model.tokens = True
    B = 8
    batch = {"rgb": torch.rand(B, 3, 224, 224, device=dev), "text": torch.randint(0, 50, (B, 77), device=dev),
             "example_ids": {"rgb": _ids(B, dev), "text": _ids(B, dev)}}
    loss = task.training_step(batch, 0)
    loss.backward()
    assert torch.isfinite(loss) and "train/ijepa_loss" in task.logged
    task.on_before_zero_grad(None)
    assert task.auxiliary_tasks["ijepa"].target_encoder.num_updates == 1


def test_fused_encoder_training_trajectory_matches_stock():
    """End-to-end regression for the whole encoder-side stack: 10 SGD steps of the bench's task (small encoders, dropout
    off) with every fusion on follow the stock-module run -- same losses within bf16 noise, same weight updates."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from mmlearn_amd import ContrastiveLoss

    dev = torch.device("cuda", 0)
    curves, deltas = [], []
    for fused in (False, True):
        task = bench.build_task(ContrastiveLoss(static_shapes=True), small=True, fused=fused).to(dev)
        task.eval()  # dropout off (BERT's default 0.1 would make the two runs draw different masks); grads still flow
        init = torch.cat([p.detach().float().flatten() for p in task.parameters()])
        opt = torch.optim.SGD(task.parameters(), lr=0.5)   # plain SGD: the weight change is the sum of the gradients
        batch = bench.synthetic_batch(1024, 0, dev)
        losses = []
        for _ in range(10):
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = task.training_step(batch, 0)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach().float().item()))
        curves.append(losses)
        deltas.append(torch.cat([p.detach().float().flatten() for p in task.parameters()]) - init)
    a, b = curves
    for x, y in zip(a, b):
        assert abs(x - y) <= 2e-2 * max(1.0, abs(x)), (a, b)
    da, db = deltas
    assert da.norm() > 0
    cos = torch.dot(da, db) / (da.norm() * db.norm())
    assert cos >= 0.98 and abs(da.norm() / db.norm() - 1) <= 0.05, (cos.item(), (da.norm() / db.norm()).item())


def test_concurrent_encoder_streams_give_the_same_step():
    """``task.concurrent_encoders = True`` (one HIP stream per tower, forward and backward) is a scheduling change only:
    same loss and gradients as the single-stream step, over several steps with optimizer updates in between (a missing
    stream dependency would show up as stale or torn tensors)."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from mmlearn_amd import ContrastiveLoss

    dev = torch.device("cuda", 0)
    runs = []
    for streams in (False, False, True):   # two single-stream runs give the run-to-run noise floor (stream-K GEMMs, f32 atomics)
        task = bench.build_task(ContrastiveLoss(), small=True, fused=True).to(dev)
        task.eval()   # dropout off: all runs must see the same function
        task.concurrent_encoders = streams
        opt = torch.optim.SGD(task.parameters(), lr=0.1)
        batch = bench.synthetic_batch(1024, 0, dev)
        rec = []
        for _ in range(4):
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = task.training_step(batch, 0)
            loss.backward()
            grads = torch.cat([p.grad.detach().float().flatten() for p in task.parameters() if p.grad is not None])
            opt.step()
            rec.append((loss.detach().float().clone(), grads))
        runs.append(rec)
        if streams:
            assert len(task._side_streams) == 1
    for k, ((l0, g0), (l1, g1), (l2, g2)) in enumerate(zip(*runs)):
        noise = (g0 - g1).abs().max().item()
        diff = (g0 - g2).abs().max().item()
        assert torch.allclose(l0, l2, rtol=1e-4, atol=1e-6), (l0.item(), l2.item())
        if k == 0:   # same weights, same inputs: equal up to the reordering of f32 atomics
            assert diff <= 1e-5 * g0.abs().max().item(), (diff, noise)
        else:        # after updates the runs drift apart chaotically -- the streamed run no faster than a repeat of the plain one
            cos = torch.dot(g0, g2) / (g0.norm() * g2.norm())
            cos_noise = torch.dot(g0, g1) / (g0.norm() * g1.norm())
            assert diff <= max(4 * noise, 1e-5 * g0.abs().max().item()) and cos >= cos_noise - 2e-4, (k, diff, noise, cos.item(), cos_noise.item())


def test_three_towers_on_three_streams_with_a_shared_head():
    """BASELINE configs[3] shape of the task (three modalities, one shared projection head, three weighted pairs) with
    ``concurrent_encoders``: the second and third tower on the side stream (``max_side_streams`` = 1 by default; the test name
    dates from one stream per tower), the shared head's parameters used from all three; same loss and gradients
    as the single-stream step, and the matcher answers for all three pairs are prefetched."""
    from mmlearn_amd import ContrastiveLoss
    from mmlearn_amd.tasks import ContrastivePretraining, LossPairSpec, ModuleKeySpec

    dev = _dev()
    D, B = 64, 512

    class Enc(torch.nn.Module):
        def __init__(self, key):
            super().__init__()
            self.key = key
            self.net = torch.nn.Sequential(torch.nn.Linear(96, 256), torch.nn.GELU(), torch.nn.Linear(256, D))

        def forward(self, inputs):
            return (self.net(inputs[self.key]),)

    def build():
        torch.manual_seed(3)
        return ContrastivePretraining(
            encoders={"rgb": Enc("rgb"), "text": Enc("text"), "audio": Enc("audio")},
            heads={"shared": {"proj": torch.nn.Linear(D, D)}},
            modality_module_mapping={m: ModuleKeySpec(encoder_key=m, head_key="shared") for m in ("rgb", "text", "audio")},
            loss=ContrastiveLoss(), optimizer=partial(torch.optim.SGD, lr=0.1),
            modality_loss_pairs=[LossPairSpec(("rgb", "text")), LossPairSpec(("rgb", "audio"), 0.5), LossPairSpec(("text", "audio"), 0.25)],
            compute_validation_loss=True, compute_test_loss=False).to(dev)

    g = torch.Generator().manual_seed(8)
    batch = {m: torch.randn(B, 96, generator=g).to(dev) for m in ("rgb", "text", "audio")}
    perm = torch.randperm(B, generator=g)
    batch["example_ids"] = {"rgb": _ids(B, dev), "text": _ids(B, dev), "audio": _ids(B, dev)[perm.to(dev)]}
    batch["audio"] = batch["audio"][perm.to(dev)]     # audio rows arrive in another order: the matcher has real work
    results = []
    for streams in (False, True):
        task = build()
        task.concurrent_encoders = streams
        recs = []
        opt = task.configure_optimizers()
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            loss = task.training_step(batch, 0)
            loss.backward()
            recs.append((loss.detach().clone(), torch.cat([p.grad.flatten() for p in task.parameters() if p.grad is not None]).clone()))
            opt.step()
        with torch.no_grad():   # the no-grad loss path (validation) under the same scheduling
            recs.append((task.validation_step(batch, 0).detach().clone(), recs[-1][1]))
        results.append(recs)
        if streams:
            assert len(task._side_streams) == 1 and task.loss_fn.prefetched_matches_used == 12
    for (l0, g0), (l1, g1) in zip(*results):
        assert torch.allclose(l0, l1, rtol=1e-6, atol=1e-7), (l0.item(), l1.item())
        assert (g0 - g1).abs().max().item() <= 1e-5 * g0.abs().max().item()


def test_hip_adamw_matches_torch_adamw():
    from mmlearn_amd.optim import AdamW

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    shapes = [(300, 768), (768,), (5000, 64), (7,), (4096 * 3 + 5,)]
    ref_p = [torch.nn.Parameter(torch.randn(*s, device=dev)) for s in shapes]
    hip_p = [torch.nn.Parameter(p.detach().clone()) for p in ref_p]
    kw = dict(lr=3e-3, betas=(0.9, 0.98), eps=1e-6)
    ref = torch.optim.AdamW([{"params": ref_p[:3], "weight_decay": 0.1}, {"params": ref_p[3:], "weight_decay": 0.0}], **kw)
    hip = AdamW([{"params": hip_p[:3], "weight_decay": 0.1}, {"params": hip_p[3:], "weight_decay": 0.0}], **kw)
    for it in range(5):
        for a, b in zip(ref_p, hip_p):
            g = torch.randn_like(a)
            a.grad = g.clone()
            b.grad = g.to(torch.bfloat16).float() if it == 2 else g.clone()   # any f32 gradient values
            if it == 2:
                a.grad = b.grad.clone()
        ref.step()
        hip.step()
        for a, b in zip(ref_p, hip_p):
            assert (a - b).abs().max() <= 2e-6 * max(1.0, a.abs().max().item()), it
    sd = hip.state_dict()
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"}
    # mid-run resume: load_state_dict replaces the moment tensors (new storages, the old ones are freed); the device pointer
    # table must follow them.  Restore BOTH optimizers from their own snapshots, poison the allocator's free list so a stale
    # pointer would show, and keep stepping.
    import copy
    ref_sd, hip_sd = copy.deepcopy(ref.state_dict()), copy.deepcopy(hip.state_dict())
    old_ptrs = {hip.state[p]["exp_avg"].data_ptr() for p in hip_p}
    ref.load_state_dict(ref_sd)
    hip.load_state_dict(hip_sd)
    assert not (old_ptrs & {hip.state[p]["exp_avg"].data_ptr() for p in hip_p})
    del ref_sd, hip_sd, sd
    junk = [torch.full((s.numel() if hasattr(s, "numel") else 1,), float("nan"), device=dev) for s in hip_p]   # may reuse the freed moment blocks
    for it in range(3):
        for a, b in zip(ref_p, hip_p):
            g = torch.randn_like(a)
            a.grad, b.grad = g.clone(), g.clone()
        ref.step()
        hip.step()
        for a, b in zip(ref_p, hip_p):
            assert torch.isfinite(b).all()
            assert (a - b).abs().max() <= 2e-6 * max(1.0, a.abs().max().item()), ("after load_state_dict", it)
        for a, b in zip(ref_p, hip_p):
            assert (ref.state[a]["exp_avg"] - hip.state[b]["exp_avg"]).abs().max() <= 1e-6
    del junk


def test_ijepa_vit_step_with_fused_blocks_matches_stock_blocks():
    """The I-JEPA task over a timm-style ViT + block predictor (tools/bench_ijepa_step.py, small sizes): with
    ``accelerate_encoder`` on encoder and predictor the first training steps give the same losses as the stock blocks
    (same seeds, hence same masks), and the EMA target stays a copy of the student."""
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_ijepa_step", os.path.join(root, "tools", "bench_ijepa_step.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    dev = _dev()
    imgs = torch.rand(32, 3, 224, 224, generator=torch.Generator().manual_seed(1)).to(dev)
    curves = []
    for fused in (False, True):
        task = tool.build(True, fused, dev)
        opt = task.configure_optimizers()
        opt = opt["optimizer"] if isinstance(opt, dict) else opt
        torch.manual_seed(7)
        losses = []
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = task.training_step({"rgb": imgs}, 0)
            loss.backward()
            opt.step()
            task.on_before_zero_grad(opt)
            losses.append(float(loss.detach().float()))
        curves.append(losses)
        for (k, t), (_, s) in zip(task.target_encoder.model.state_dict().items(), task.encoder.state_dict().items()):
            assert torch.equal(t, s), k
    for a, b in zip(*curves):
        assert abs(a - b) <= 1e-2 * max(1.0, abs(a)), curves


def test_ijepa_step_makes_no_host_synchronisation():
    """the whole I-JEPA training step -- teacher forward, context-encoder mask branch (vision.py:335-337,
    here ``ops.apply_masks`` on the host-built indices that ride on the mask list), predictor assembly, fused target /
    loss kernel, backward, AdamW, EMA update -- runs under ``torch.cuda.set_sync_debug_mode("error")``: any
    ``.item()`` / ``nonzero`` / blocking copy on the way raises.  (The reference syncs once per mask in boolean-mask
    indexing, masking.py:264-283.)  Also: the drop-in ``apply_masks`` with IndexedMasks equals boolean indexing."""
    import importlib.util, os
    from mmlearn_amd import IndexedMasks, apply_masks
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_ijepa_step", os.path.join(root, "tools", "bench_ijepa_step.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    dev = _dev()
    imgs = torch.rand(16, 3, 224, 224, generator=torch.Generator().manual_seed(1)).to(dev)
    task = tool.build(True, True, dev)
    opt = task.configure_optimizers()
    opt = opt["optimizer"] if isinstance(opt, dict) else opt

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = task.training_step({"rgb": imgs}, 0)
        loss.backward()
        opt.step()
        task.on_before_zero_grad(opt)
        return loss

    torch.manual_seed(7)
    step()                      # first step: lazy one-time set-up (pointer tables, workspaces) may read back
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        losses = [step() for _ in range(3)]
    finally:
        torch.cuda.set_sync_debug_mode("default")
    vals = [float(l.detach().float()) for l in losses]
    assert all(v == v and v > 0 for v in vals), vals

    # drop-in apply_masks: indices attached to the mask list == boolean-mask indexing of every mask
    x = torch.randn(4, 196, 32, device=dev)
    info = task.mask_generator(batch_size=4)
    masks = IndexedMasks([m.to(dev) for m in info["predictor_masks"]], info["predictor_indices"].to(dev))
    torch.cuda.set_sync_debug_mode("error")
    try:
        got = apply_masks(x, masks)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    ref = torch.cat([x[m.bool()].view(4, -1, 32) for m in masks], 0)
    assert torch.equal(got, ref)
    with pytest.raises(ValueError):
        IndexedMasks(list(masks), info["predictor_indices"][:2].to(dev))
