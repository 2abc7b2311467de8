"""CPU tests of the host-side logic: mask generator RNG sequence (golden G7), modality registry,
registry wiring, task construction contracts."""

import numpy as np
import pytest
import torch

from conftest import Golden

MASKS = Golden("g7_masks")


@pytest.mark.parametrize("name", MASKS.names())
def test_mask_generator_vs_reference(name):
    from mmlearn_amd.masking import IJEPAMaskGenerator

    c = MASKS[name]
    if name.startswith("seed"):
        seed, b = name[4:].split("_b")
        torch.manual_seed(int(seed))
        b = int(b)
        m = IJEPAMaskGenerator()(batch_size=b)
    else:
        torch.manual_seed(4)
        b = 2
        m = IJEPAMaskGenerator(input_size=(96, 128), patch_size=8, npred=2, nenc=2)(batch_size=b)
    after = torch.randint(0, 2**31, (1,)).item()
    np.testing.assert_array_equal(torch.stack(m["encoder_masks"]).numpy(), c["enc"])
    np.testing.assert_array_equal(torch.stack(m["predictor_masks"]).numpy(), c["pred"])
    assert after == int(c["rng_after"])  # the global RNG advanced exactly as in the reference
    # the index form equals nonzero() of the mask form
    for masks, idx in ((m["encoder_masks"], m["encoder_indices"]), (m["predictor_masks"], m["predictor_indices"])):
        for k, mk in enumerate(masks):
            assert mk.dtype == torch.int32 and mk.shape[0] == b
            np.testing.assert_array_equal(idx[k, 0].numpy(), np.nonzero(mk[0].numpy())[0])


def test_masks_to_indices_host_path():
    from mmlearn_amd import ops

    m = torch.zeros(3, 20, dtype=torch.int32)
    m[:, [2, 5, 7]] = 1
    idx = ops.masks_to_indices([m, m[0]], 3, torch.device("cpu"))
    assert idx.shape == (2, 1, 3) and idx.dtype == torch.int32
    assert idx[0, 0].tolist() == [2, 5, 7]
    per = torch.zeros(3, 20, dtype=torch.int32)
    per[0, [1, 2]] = 1; per[1, [3, 4]] = 1; per[2, [5, 19]] = 1
    assert ops.masks_to_indices([per], 3, torch.device("cpu"))[0].tolist() == [[1, 2], [3, 4], [5, 19]]
    bad = per.clone(); bad[0, 9] = 1
    with pytest.raises(ValueError):
        ops.masks_to_indices([bad], 3, torch.device("cpu"))
    with pytest.raises(ValueError):
        ops.masks_to_indices([m[:2]], 3, torch.device("cpu"))


def test_modalities_and_registry():
    from mmlearn_amd import Modalities, registry

    rgb = Modalities.get_modality("RGB")
    assert (rgb.name, rgb.embedding, rgb.mask, rgb.target) == ("rgb", "rgb_embedding", "rgb_mask", "rgb_target")
    assert Modalities.has_modality("audio") and not Modalities.has_modality("smell")
    assert Modalities.rgb is rgb
    import mmlearn_amd.tasks  # noqa: F401

    keys = set(registry.REGISTERED)
    assert {("modules/losses", "ContrastiveLossHIP"), ("task", "ContrastivePretrainingHIP"), ("task", "IJEPAHIP")} <= keys


def test_loss_constructor_contract():
    from mmlearn_amd import ContrastiveLoss

    l = ContrastiveLoss(l2_normalize=True, local_loss=True, gather_with_grad=True, cache_labels=True)
    assert (l.l2_normalize, l.local_loss, l.gather_with_grad, l.cache_labels) == (True, True, True, True)
    assert ContrastiveLoss(modality_alignment=True).modality_alignment is True
    with pytest.raises(ValueError):
        ContrastiveLoss(compute_dtype=torch.float16)


def test_ema_requires_gpu_and_configure():
    from mmlearn_amd import ExponentialMovingAverage

    net = torch.nn.Linear(3, 2)
    ema = ExponentialMovingAverage(net, 0.9, 1.0, 10)
    with pytest.raises(RuntimeError, match="not configured"):
        ema.step(net)
    ema.configure_model("cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ema.step(net)
    assert ExponentialMovingAverage.get_annealed_rate(0.996, 1.0, 500, 1000) == pytest.approx(0.998)


def test_training_task_optimizer_split():
    from functools import partial

    from mmlearn_amd.tasks import TrainingTask

    class T(TrainingTask):
        def __init__(self, **kw):
            super().__init__(**kw)
            self.net = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.LayerNorm(4))

    t = T(optimizer=partial(torch.optim.AdamW, lr=1e-3, weight_decay=0.2), loss_fn=lambda a, b: a,
          lr_scheduler={"scheduler": partial(torch.optim.lr_scheduler.StepLR, step_size=1), "extras": {"interval": "step"}})
    out = t.configure_optimizers()
    groups = {g["name"]: g for g in out["optimizer"].param_groups}
    assert groups["weight_decay_params"]["weight_decay"] == 0.2 and len(groups["weight_decay_params"]["params"]) == 1
    assert groups["no_weight_decay_params"]["weight_decay"] == 0.0 and len(groups["no_weight_decay_params"]["params"]) == 3
    assert out["lr_scheduler"]["interval"] == "step"
    with pytest.raises(ValueError):
        T()
    with pytest.warns(UserWarning):
        assert T(loss_fn=lambda a, b: a).configure_optimizers() is None


def test_tools_and_entry_points_parse_and_bench_launches_its_own_ranks():
    """Every script shipped next to the package is at least syntactically valid (they are run by hand on the GPU box).
    ``python bench.py --gpus 2`` with no launcher starts its own two rank processes before any GPU call: on this GPU-less
    box the RANKS fail, each with a clear "no GPU" message, and the parent relays a non-zero exit code; a --gpus that
    disagrees with a launcher's WORLD_SIZE is still refused."""
    import ast, glob, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = glob.glob(os.path.join(root, "tools", "*.py")) + [os.path.join(root, f) for f in ("bench.py", "__graft_entry__.py")]
    assert len(files) >= 10
    for f in files:
        ast.parse(open(f).read(), filename=f)
    if torch.cuda.is_available():
        return   # on a GPU box the ranks would really run: the launch path is exercised by the driver there
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0
    assert "rank 0: no GPU visible" in r.stderr and "rank 1: no GPU visible" in r.stderr, r.stderr[-2000:]
    assert "stopping the other ranks" in r.stderr or "exited with code" in r.stderr
    assert r.stdout.strip() == ""     # no result line from a failed run
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=4" in (r.stderr + r.stdout)


def test_tower_ddp_strategy_hands_the_wrapping_to_the_task():
    """``trainer.strategy: tower_ddp`` (mmlearn/cli/run.py:52-61 builds the Trainer from YAML): a task with
    ``concurrent_encoders`` gets ``wrap_towers_in_ddp`` with the strategy's DDP kwargs and NO outer DDP (the model is
    returned as is, so Lightning's ``Strategy.training_step`` calls the module directly); any other module falls through
    to the base class; the hardware-queue setting is exported before any HIP call."""
    import os
    from mmlearn_amd import strategy as S

    os.environ.pop("GPU_MAX_HW_QUEUES", None)
    st = S.TowerDDPStrategy(bucket_cap_mb=50)
    assert os.environ["GPU_MAX_HW_QUEUES"] == "8"
    calls = []

    class Task(torch.nn.Module):
        concurrent_encoders = True

        def wrap_towers_in_ddp(self, **kw):
            calls.append(kw)

    class Wrapper(torch.nn.Module):   # Lightning's forward-redirection wrappers keep the module in ``_forward_module``
        def __init__(self, m):
            super().__init__()
            self._forward_module = m

    t = Task()
    assert st._setup_model(t) is t and calls == [{"bucket_cap_mb": 50, "gradient_as_bucket_view": True}]
    st._register_ddp_hooks()     # no outer DDP instance: nothing to register, must not assert
    w = Wrapper(Task())
    assert S.TowerDDPStrategy()._setup_model(w) is w and len(calls) == 2
    # a task that does not overlap its towers, or any plain module: stock DDP path of the base class
    base_calls = []
    orig = S.DDPStrategy._setup_model
    S.DDPStrategy._setup_model = lambda self, m: base_calls.append(m) or "ddp"
    try:
        t2 = Task()
        t2.concurrent_encoders = False
        os.environ.pop("GPU_MAX_HW_QUEUES", None)
        keep = S.TowerDDPStrategy(concurrent_encoders=False)          # keeps the task's own setting ...
        assert "GPU_MAX_HW_QUEUES" not in os.environ                  # ... and leaves the process environment alone
        assert keep._setup_model(t2) == "ddp" and S.TowerDDPStrategy()._setup_model(torch.nn.Linear(2, 2)) == "ddp"
    finally:
        S.DDPStrategy._setup_model = orig
    assert len(base_calls) == 2 and len(calls) == 2
    # choosing the strategy IS the request: a task built by hydra with the default concurrent_encoders=False is switched on
    t3 = Task()
    t3.concurrent_encoders = False
    with pytest.warns(RuntimeWarning, match="switches task.concurrent_encoders on"):       # ... and says so
        assert S.TowerDDPStrategy()._setup_model(t3) is t3 and t3.concurrent_encoders is True and len(calls) == 3


def test_per_tower_ddp_keeps_the_reference_checkpoint_keys():
    """``wrap_towers_in_ddp`` used to put the DistributedDataParallel instances INTO ``encoders`` / ``heads``, so
    ``state_dict()`` grew ``encoders.<m>.module.*`` keys -- not the reference's format, and a checkpoint written by such a run
    did not load back (Lightning restores before the strategy wraps).  The wrappers now live outside the module tree."""
    import os

    import torch.distributed as dist

    import tiny_models
    from mmlearn_amd import ContrastiveLoss
    from mmlearn_amd.tasks import ContrastivePretraining

    def build():
        torch.manual_seed(3)
        enc = {"rgb": tiny_models.FlatMLPEncoder("rgb", 3 * 4 * 4, 8, 6), "text": tiny_models.TokenMLPEncoder("text", 20, 5, 6)}
        heads = {"rgb": torch.nn.Linear(6, 6), "text": torch.nn.Linear(6, 6)}
        return ContrastivePretraining(encoders=enc, heads=heads, loss=ContrastiveLoss(), concurrent_encoders=False, max_side_streams=2)

    plain = build()
    assert plain.max_side_streams == 2 and plain.match_ahead is True          # constructor keywords (YAML-reachable)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(29800 + os.getpid() % 100)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        task = build()
        keys_before = list(task.state_dict().keys())
        task.wrap_towers_in_ddp()
        assert len(task._tower_ddp) == 4 and list(task.state_dict().keys()) == keys_before
        assert not any(".module." in k for k in task.state_dict())
        with torch.no_grad():
            for p_ in task.parameters():
                p_.add_(1.0)
        sd = task.state_dict()
        missing, unexpected = plain.load_state_dict(sd, strict=True)            # wrapped -> unwrapped
        assert not missing and not unexpected
        task2 = build()
        task2.wrap_towers_in_ddp()
        task2.load_state_dict(plain.state_dict(), strict=True)                   # unwrapped -> wrapped
        from mmlearn_amd.modalities import Modalities
        x = {"rgb": torch.rand(2, 3, 4, 4)}
        out_w = task2.encode(x, Modalities.get_modality("rgb"))                  # runs through the DDP instance
        out_p = plain.encode(x, Modalities.get_modality("rgb"))
        assert torch.allclose(out_w, out_p)
        out_w.sum().backward()                                                   # DDP's reducer sees the backward
        assert all(p_.grad is not None for p_ in task2.encoders["rgb"].parameters())
        # the wrappers sit outside the module tree, so the task looks after them itself --
        import copy
        import pickle
        clone = copy.deepcopy(task2)                                             # a copy starts without process-group-bound wrappers
        assert "_tower_ddp" not in clone.__dict__ and torch.allclose(clone.encode(x, Modalities.get_modality("rgb")), out_p)
        assert "_tower_ddp" not in pickle.loads(pickle.dumps(task2.__getstate__())) and "_tower_ddp" in task2.__dict__
        task2.eval()                                                             # train() / eval() reach them through encode()
        task2.encode(x, Modalities.get_modality("rgb"))
        assert not task2._tower_ddp[("encoders", "rgb")].training
        task2.train()
        task2.encode(x, Modalities.get_modality("rgb"))
        assert task2._tower_ddp[("encoders", "rgb")].training
        task2.encoders["rgb"] = tiny_models.FlatMLPEncoder("rgb", 3 * 4 * 4, 8, 6)   # a tower replaced after wrapping is refused, not ignored
        with pytest.raises(RuntimeError, match="replaced after wrap_towers_in_ddp"):
            task2.encode(x, Modalities.get_modality("rgb"))
    finally:
        dist.destroy_process_group()


def test_tuned_gemm_selection_file_is_well_formed():
    """mmlearn_amd/tuned/gemm_gfx950.csv: TunableOp validators first, then one plain / bias GEMM entry per line (no strided-batched
    entries -- a library candidate for HTSAT's 24-wide batched products faults --, every dimension a multiple of 16: the towers'
    Linear shapes, from HTSAT's 16-channel patch embedding and 96-wide stage up); enable() has no CPU path."""
    import re

    import pytest
    import torch

    from mmlearn_amd import tuned

    lines = open(tuned.DEFAULT_FILE).read().splitlines()
    vals = [ln for ln in lines if ln.startswith("Validator,")]
    assert {ln.split(",")[1] for ln in vals} >= {"PT_VERSION", "HIPBLASLT_VERSION", "ROCBLAS_VERSION", "GCN_ARCH_NAME"}
    assert any("gfx950" in ln for ln in vals)
    body = lines[len(vals):]
    assert len(body) >= 10 and len({tuple(ln.split(",")[:2]) for ln in body}) == len(body)
    for ln in body:
        op, key, sol, ms = ln.split(",")
        assert op.split("_")[0] in ("GemmTunableOp", "GemmAndBiasTunableOp") and "BFloat16" in op
        m = re.match(r"(tn|nt|nn)_(\d+)_(\d+)_(\d+)_ld_", key)
        assert "Batched" not in op and m and all(int(d) % 16 == 0 for d in m.groups()[1:])
        assert sol == "Default" or sol.startswith(("Gemm_Hipblaslt_", "Gemm_Rocblas_"))
        assert float(ms) > 0
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            tuned.enable()


def test_round4_host_side_planning_functions_without_a_gpu():
    """Pure host arithmetic of the round-4 entry points (no compute call): scratch sizes of the sorted-run embedding backward follow
    its level structure, the windowed-attention launch has a pair count that is a multiple of 8 whenever the batch allows (the XCD map
    of the block ids relies on it) and never more slices than samples, the MLP node pads row counts to whole 256-row tiles."""
    from mmlearn_amd import _lib, fused

    lib = _lib.lib()
    d = 768
    assert lib.mmk_embedding_bwd_scratch_bytes(1024, d) == 256                      # one level: nothing but the alignment slack
    n1 = 2 * ((78848 + 31) // 32)
    n2 = 2 * ((n1 + 31) // 32)
    want = 0
    for n in (n1, n2):                                                                  # 78,848 rows -> 4,928 -> 308 entries
        want = (want + n * d * 4 + n * 8 + 255) // 256 * 256
    assert n2 <= 1024 < n1 and lib.mmk_embedding_bwd_scratch_bytes(78848, d) == want + 256
    for B, nW, H in ((256, 64, 4), (256, 16, 8), (256, 4, 16), (256, 1, 32), (3, 64, 2), (2, 1, 16), (1, 1, 1)):
        blocks = lib.mmk_win_attn_blocks(B, nW, H)
        assert blocks % (nW * H) == 0
        nsplit = blocks // (nW * H)
        assert 1 <= nsplit <= B and (B < 8 or (nW * nsplit) % 8 == 0), (B, nW, H, nsplit)
    assert lib.mmk_win_attn_supported(64, 24, 96) == 1 and lib.mmk_win_attn_supported(64, 32, 96) == 1
    assert lib.mmk_win_attn_supported(49, 32, 96) == 0 and lib.mmk_win_attn_supported(64, 64, 768) == 0
    assert lib.mmk_colsum_rows_slices(1) == 1 and lib.mmk_colsum_rows_slices(1 << 20) == 1024
    assert [fused._pad_rows(m) for m in (256, 1000, 2048, 2049, 5760, 25088)] == [256, 1000, 2048, 2304, 5888, 25088]


def test_window_attention_patch_takes_only_the_forward_signatures_it_mirrors():
    """``fuse_window_attention`` on CPU-built HF models: all of HTSAT's attention modules and layers (``ClapAudioSelfAttention.forward(hidden_states,
    attention_mask, output_attentions)``), none of a Swin whose forwards carry further arguments; parameters and state_dict keys untouched."""
    from transformers import ClapAudioConfig, ClapAudioModel

    from mmlearn_amd import fused

    m = ClapAudioModel(ClapAudioConfig(depths=(2, 1), num_attention_heads=(2, 4), patch_embeds_hidden_size=48, hidden_size=96))
    keys = list(m.state_dict().keys())
    assert fused.fuse_window_attention(m) == 3 and fused.fuse_window_attention(m) == 0          # idempotent
    assert sum(getattr(getattr(x, "forward", None), "__func__", None) is fused._swin_layer_forward for x in m.modules()) == 3
    assert list(m.state_dict().keys()) == keys
    try:
        from transformers import SwinConfig, SwinModel
    except ImportError:
        return
    sw = SwinModel(SwinConfig(image_size=64, embed_dim=48, depths=(1, 1), num_heads=(2, 4), window_size=8))
    import inspect
    takes = [p for p in inspect.signature(next(x for x in sw.modules() if hasattr(x, "relative_position_bias_table")).forward).parameters]
    assert fused.fuse_window_attention(sw) == (len([x for x in sw.modules() if hasattr(x, "relative_position_bias_table")])
                                               if takes == ["hidden_states", "attention_mask", "output_attentions"] else 0)



def test_measurement_seams_are_scoped():
    """VERDICT r4 housekeeping: the module-level seams that tests and tools flip (`kernels.FUSED_LOSS`, `TN_MIN_ROWS`, ...,
    `ContrastiveLoss._force_gather`) sit behind context managers that restore the previous value -- also when the block raises."""
    import mmlearn_amd.kernels as K
    from mmlearn_amd import ContrastiveLoss

    before = (K.FUSED_LOSS, K.TN_MIN_ROWS, K.PAIR_MIRRORS, K.BOUNDED_SOFTMAX)
    with pytest.raises(ZeroDivisionError):
        with K.seams(FUSED_LOSS=False, TN_MIN_ROWS=256):
            assert (K.FUSED_LOSS, K.TN_MIN_ROWS) == (False, 256)
            with K.seams(PAIR_MIRRORS=False):   # nests (re-entrant lock)
                assert K.PAIR_MIRRORS is False
            assert K.PAIR_MIRRORS is True
            1 / 0
    assert (K.FUSED_LOSS, K.TN_MIN_ROWS, K.PAIR_MIRRORS, K.BOUNDED_SOFTMAX) == before
    with pytest.raises(KeyError):
        with K.seams(FUSED=False):
            pass
    fn = ContrastiveLoss()
    with fn.forcing_gather():
        assert fn._force_gather
    assert not fn._force_gather


def test_the_step_traces_with_fullgraph_through_the_custom_ops(monkeypatch):
    """mmlearn_amd/compiled.py on the CPU: ``torch.compile(task.training_step, backend="aot_eager", fullgraph=True)`` traces the whole
    step -- encoders, the L2-normalise operator, the loss operator, logging -- without a graph break, and AOT autograd routes the
    backward through the registered backward operators.  The two kernel layers behind the operators are replaced by CPU stand-ins
    here (the operators themselves, their fake implementations and the token registry are the product's); the GPU twin of this test
    compares compiled and eager steps bit for bit (tests/test_graph_capture_gpu.py)."""
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import graph_step as G
    from mmlearn_amd import compiled, kernels as K, losses

    def l2f(x, twin=False):
        n = x.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        return (x / n).contiguous(), (1 / n).reshape(-1).float()

    def l2b(x, dy, inv):
        y = x * inv.view(-1, 1)
        return (dy - y * (dy * y).sum(-1, keepdim=True)) * inv.view(-1, 1)

    class Run:   # the loss in closed form (2 modalities, rows paired by position)
        def __init__(self, module, embs, ids, scale, pairs, fully_paired):
            self.embs, self.scale, self.pairs = embs, scale, pairs
            assert list(ids) == ["rgb", "text"] and fully_paired is True and pairs[0].modalities == ("rgb", "text") and pairs[0].weight == 1.0

        def forward(self):
            a, b = (t.detach().float() for t in self.embs.values())
            lg = self.scale.detach().float() * a @ b.t()
            lab = torch.arange(a.shape[0])
            self.saved = (a, b, lg)
            return (torch.nn.functional.cross_entropy(lg, lab) + torch.nn.functional.cross_entropy(lg.t(), lab)) / 2

        def backward(self, g):
            a, b, lg = self.saved
            n, s = a.shape[0], self.scale.detach().float()
            G_ = (torch.softmax(lg, 1) + torch.softmax(lg, 0) - 2 * torch.eye(n)) / (2 * n) * g
            return (G_ * (a @ b.t())).sum(), [s * G_ @ b, s * G_.t() @ a]

    monkeypatch.setattr(K, "require_gpu", lambda *a, **k: None)
    monkeypatch.setattr(K, "l2norm_fwd", l2f)
    monkeypatch.setattr(K, "l2norm_bwd", l2b)
    monkeypatch.setattr(losses, "_Run", Run)
    dev = torch.device("cpu")
    batch = G.make_batch(16, dev)
    results = []
    for traced in (False, True):
        torch._dynamo.reset()
        torch.manual_seed(0)
        task, _ = G.make(dev, 64)
        opt = torch.optim.SGD(task.parameters(), lr=0.1)
        step = torch.compile(task.training_step, backend="aot_eager", fullgraph=True) if traced else None
        for _ in range(2):
            opt.zero_grad(set_to_none=False)
            if traced:
                loss = step(batch, 0)
            else:   # the eager reference of the same closed form (plain autograd)
                out = task(batch)
                with torch.no_grad():
                    task.log_logit_scale.clamp_(0, 4.6052)
                a, b = out["rgb_embedding"], out["text_embedding"]
                lg = task.log_logit_scale.exp() * a @ b.t()
                lab = torch.arange(16)
                loss = (torch.nn.functional.cross_entropy(lg, lab) + torch.nn.functional.cross_entropy(lg.t(), lab)) / 2
            loss.backward()
            opt.step()
        results.append([p.detach().clone() for p in task.parameters()])
        if traced:
            assert set(task.logged) == {"train/loss", "train/logit_scale"} and not compiled._RUNS   # every parked run was redeemed
    torch._dynamo.reset()
    for e, c in zip(*results):
        assert torch.allclose(e, c, rtol=1e-4, atol=1e-6)


def test_compiled_loss_meta_round_trip_and_run_registry():
    """mmlearn_amd/compiled.py host logic: everything of a loss call that is not a tensor travels as one string operand and comes
    back unchanged; parked runs are redeemed once, and forward passes whose backward never comes are evicted oldest first."""
    from mmlearn_amd import compiled
    from mmlearn_amd.tasks import LossPairSpec

    pairs = [LossPairSpec(("rgb", "text")), LossPairSpec(("rgb", "audio"), 0.5), LossPairSpec(("text", "audio"), 1e-3)]
    for fp in (True, False, None):
        meta = compiled._encode_meta(["rgb_embedding", "text_embedding", "audio_embedding"], ["rgb", "text", "audio"], pairs, fp)
        ek, ik, ps, got_fp = compiled._decode_meta(meta)
        assert ek == ["rgb_embedding", "text_embedding", "audio_embedding"] and ik == ["rgb", "text", "audio"] and got_fp is fp
        assert [(p.modalities, p.weight) for p in ps] == [(p.modalities, p.weight) for p in pairs]
    assert compiled._decode_meta(compiled._encode_meta(["a_embedding"], [], [], None))[1:3] == ([], [])
    compiled._RUNS.clear()
    toks = [compiled._park(object()) for _ in range(compiled.MAX_PARKED_RUNS + 3)]
    assert len(compiled._RUNS) == compiled.MAX_PARKED_RUNS and toks[0] not in compiled._RUNS and toks[-1] in compiled._RUNS
    compiled._RUNS.clear()
    # every loss module is its own call site, copies included
    import copy
    from mmlearn_amd import ContrastiveLoss
    a = ContrastiveLoss()
    b = copy.deepcopy(a)
    assert a._site_id != b._site_id and compiled._SITES[a._site_id] is a and compiled._SITES[b._site_id] is b


def test_key_mask_recognition_is_by_shape_and_strides_only():
    """attention.key_mask_view (the CPU-checkable half of the mask support): what PROVES a key-padding mask is accepted -- [B, L],
    [B, 1, 1, L], stride-0 expansions over heads / queries -- and a materialised [B, 1, L, L] tensor, a 3-D per-query mask, a wrong
    batch or length are refused.  (Device tensors only: the test fakes ``is_cuda`` with a subclass, the function never touches data.)"""
    from mmlearn_amd.attention import key_mask_view

    class Dev(torch.Tensor):
        is_cuda = property(lambda self: True)

    def dev(t):
        return t.as_subclass(Dev)

    B, L = 3, 7
    keep = torch.rand(B, L) > 0.3
    v, add = key_mask_view(dev(keep), B, L)
    assert add is False and tuple(v.shape) == (B, L)
    v, add = key_mask_view(dev(keep[:, None, None, :]), B, L)
    assert add is False and torch.equal(torch.Tensor(v).bool(), keep)
    v, add = key_mask_view(dev(keep[:, None, None, :].expand(B, 4, L, L)), B, L)
    assert torch.equal(torch.Tensor(v).bool(), keep)
    addm = torch.zeros(B, 1, 1, L).masked_fill(~keep[:, None, None, :], torch.finfo(torch.float32).min)
    assert key_mask_view(dev(addm), B, L)[1] is True
    assert key_mask_view(dev(keep[:, None, None, :].expand(B, 1, L, L).contiguous()), B, L) is None   # could hold anything
    assert key_mask_view(dev(keep[:, None, :]), B, L) is None                                          # 3-D: a per-query mask
    assert key_mask_view(dev(keep), B + 1, L) is None and key_mask_view(dev(keep), B, L + 1) is None
    assert key_mask_view(keep, B, L) is None                                                           # host tensors are not served
    assert key_mask_view(dev(keep.half()[:, None, None, :]), B, L) is None                             # fp16 masks: not a served dtype


def test_mask_scope_leaves_hosts_and_checkpointed_models_to_hf():
    """fused.scope_key_masks on the CPU: the scope is installed on the innermost HF text model only, copies get their own scope, and the
    pre-hook does nothing for host tensors (so a CPU forward with a padding mask still equals the stock model)."""
    import copy
    from transformers import BertConfig, BertModel

    from mmlearn_amd import fused

    torch.manual_seed(0)
    cfg = BertConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=1, intermediate_size=128, vocab_size=100,
                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    stock = BertModel(cfg, add_pooling_layer=False).eval()
    wrapped = torch.nn.Sequential()
    wrapped.add_module("tower", copy.deepcopy(stock))
    assert fused.fuse_qkv_attention(wrapped) == 2
    scope = wrapped.tower._mmk_mask_scope
    assert all(l.attention.self._mmk_mask_scope is scope for l in wrapped.tower.encoder.layer)
    assert fused.scope_key_masks(wrapped) == 0                      # once
    twin = copy.deepcopy(wrapped)
    assert twin.tower._mmk_mask_scope is not scope and twin.tower.encoder.layer[0].attention.self._mmk_mask_scope is twin.tower._mmk_mask_scope
    ids = torch.randint(0, 100, (3, 9))
    mask = torch.ones(3, 9, dtype=torch.long)
    mask[1, 5:] = 0
    with torch.no_grad():
        a = stock(input_ids=ids, attention_mask=mask).last_hidden_state
        b = wrapped.tower(input_ids=ids, attention_mask=mask).last_hidden_state
    assert torch.allclose(a, b, atol=1e-6) and scope.key_bias is None and scope.stripped is False
    s = fused._MaskScope()
    s.mask2d = mask
    add = s.additive(torch.float32)
    assert add.shape == (3, 1, 1, 9) and (add[1, 0, 0, 5:] == torch.finfo(torch.float32).min).all() and (add[0] == 0).all()


def test_tower_operator_host_side_round_trips_a_batch_and_keys_towers_per_module_object():
    """mmlearn_amd.compiled hands an accelerated encoder to torch.compile as ONE operator.  Host side: the (nested) batch dict is split
    into tensor operands + one string operand and rebuilt inside the operator; the registry key lives on the module object, and a
    deep copy -- which carries its original's key -- gets its own on ``mark_tower`` (the task's ``__deepcopy__`` does that)."""
    import copy

    from mmlearn_amd import compiled as C

    batch = {"rgb": torch.zeros(2, 3), "attention_mask": torch.ones(2, 5, dtype=torch.long), "fully_paired": True, "note": "a|b=c:d", "k": None, "n": 7,
             "scale": 0.25, "example_ids": {"rgb": torch.arange(4).view(2, 2), "text": torch.arange(4).view(2, 2)}, "obj": object()}
    keys, tensors, consts = C._flatten_inputs(batch)
    assert keys == ["rgb", "attention_mask", "example_ids/rgb", "example_ids/text"] and len(tensors) == 4
    assert consts == {"fully_paired": True, "note": "a|b=c:d", "k": None, "n": 7, "scale": 0.25}          # the python object stays outside
    meta = C._encode_tower_meta(keys, consts, [0])
    m = C._decode_tower_meta(meta)
    assert m == {"keys": keys, "consts": consts, "grad_inputs": [0]}
    back = C._rebuild_inputs(m["keys"], tensors, m["consts"])
    assert set(back) == set(batch) - {"obj"} and back["example_ids"]["text"] is batch["example_ids"]["text"] and back["fully_paired"] is True
    assert C._decode_tower_meta(C._encode_tower_meta([], {}, [])) == {"keys": [], "consts": {}, "grad_inputs": []}

    enc = torch.nn.Linear(3, 2)
    assert not C.is_opaque_tower(enc)
    C.mark_tower(enc)
    tid = enc._mmk_tower_id
    C.mark_tower(enc)
    assert C.is_opaque_tower(enc) and enc._mmk_tower_id == tid and C._TOWERS[tid] is enc
    twin = copy.deepcopy(enc)
    assert twin._mmk_tower_id == tid and C._TOWERS[tid] is enc      # the copy carries the key of the original ...
    C.mark_tower(twin)
    assert twin._mmk_tower_id != tid and C._TOWERS[twin._mmk_tower_id] is twin and C._TOWERS[tid] is enc   # ... until it is registered itself
