"""GPU: the configs[0]-sized training step (MLP encoders, batch 64) captured into a HIP graph -- encoders, l2-normalise, the
one-launch loss with its gradients, backward, AdamW -- replays to the same parameters as the eager step, bit for bit.
Nothing on that path may read back to the host, allocate outside the capture pool or launch on a stream the capture does not
follow; SURVEY 8(f1) lists graph capture of the step (mmlearn/cli/run.py:139 plumbs torch.compile for the same purpose)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_small_step_is_graph_capturable_and_replays_bit_identically():
    assert torch.cuda.is_available(), "needs a MI355X"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import graph_step as G
    from mmlearn_amd import kernels as K

    dev = torch.device("cuda", 0)
    batch = G.make_batch(64, dev)
    calls = []
    real = K.clip_fused_forward
    K.clip_fused_forward = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        task_e, opt_e = G.make(dev, 512)
        for _ in range(3 + 4):
            G.step(task_e, opt_e, batch)
        task_g, opt_g = G.make(dev, 512)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                G.step(task_g, opt_g, batch)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        n_before = len(calls)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            loss_g = G.step(task_g, opt_g, batch)
        assert len(calls) == n_before + 1, "the captured step must contain the one-launch loss"
        for _ in range(4):
            graph.replay()
        torch.cuda.synchronize()
    finally:
        K.clip_fused_forward = real
    assert torch.isfinite(loss_g.detach()).all()
    for pe, pg in zip(task_e.parameters(), task_g.parameters()):
        assert torch.equal(pe.detach(), pg.detach())


def _copy_batch_(dst, src):
    """Refill the captured step's input tensors in place (a graph replays on fixed addresses)."""
    for k in ("rgb", "text"):
        dst[k].copy_(src[k])
    for k in ("rgb", "text"):
        dst["example_ids"][k].copy_(src["example_ids"][k])


def test_step_with_the_id_matcher_and_the_repos_own_adamw_is_capturable():
    """capture beyond ``fully_paired``.  The text rows arrive shuffled, so the loss runs the id matcher -- inside
    the graph, with its pair count taken from the eager warm-up and verified on the device at every replay -- and the optimizer is
    ``mmlearn_amd.optim.AdamW(capturable=True)`` (step count and learning rate in device words).  Replays with NEW batches (other
    pixels, other permutations) give bit-identical parameters to the same steps launched eagerly; a batch that pairs differently
    than the captured one (a repeated id) raises the device flag and poisons the loss instead of training on a wrong pairing."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import graph_step as G

    dev = torch.device("cuda", 0)
    batches = [G.make_batch(64, dev, shuffled=True, seed=10 + k) for k in range(7)]
    # eager reference: 3 warm-up steps + 4 more, each on its own batch
    task_e, opt_e = G.make(dev, 512, own_adamw=True)
    for k in range(7):
        G.step(task_e, opt_e, batches[k])
    # captured: 3 eager warm-up steps on a side stream, then one capture, replayed on batches 3 .. 6 through a static input buffer
    task_g, opt_g = G.make(dev, 512, own_adamw=True)
    static = G.make_batch(64, dev, shuffled=True, seed=99)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for k in range(3):
            _copy_batch_(static, batches[k])
            G.step(task_g, opt_g, static)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    _copy_batch_(static, batches[3])
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loss_g = G.step(task_g, opt_g, static)
    losses = []     # capture itself does not execute: the replay for batch 3 comes first
    for k in range(3, 7):
        _copy_batch_(static, batches[k])
        graph.replay()
        losses.append(float(loss_g.detach()))
    torch.cuda.synchronize()
    assert all(l == l for l in losses), losses                      # no NaN: every replayed batch paired like the captured one
    assert not bool(task_g.loss_fn.capture_mismatch)
    for pe, pg in zip(task_e.parameters(), task_g.parameters()):
        assert torch.equal(pe.detach(), pg.detach())
    assert float(opt_g.state[next(iter(task_g.parameters()))]["step"]) == 7.0
    # a batch with a repeated id pairs differently (65 pairs, repeated rows): flagged on the device, loss poisoned
    bad = G.make_batch(64, dev, shuffled=True, seed=5)
    bad["example_ids"]["text"][7] = bad["example_ids"]["text"][9]
    _copy_batch_(static, bad)
    graph.replay()
    torch.cuda.synchronize()
    assert bool(task_g.loss_fn.capture_mismatch) and not (float(loss_g.detach()) == float(loss_g.detach()))


@pytest.mark.parametrize("backend,fullgraph,shuffled", [("eager", False, False), ("aot_eager", False, False), ("aot_eager", True, False),
                                                        ("aot_eager", True, True)])
def test_the_task_runs_under_torch_compile_and_trains_to_the_same_parameters(backend, fullgraph, shuffled):
    """The reference plumbs ``torch.compile`` at mmlearn/cli/run.py:139 (``torch.compile(task, **compile_kwargs)``).  The ops on the
    step's path are ``torch.library`` custom ops with fake implementations (mmlearn_amd/compiled.py): TorchDynamo and AOT autograd
    trace THROUGH them, so the step compiles with ``fullgraph=True`` -- no graph break at the L2-normalise kernel or anywhere in the
    loss -- also when the loss has to run the id matcher (``shuffled``).  With the Triton-free backends
    (``eager``: Dynamo only; ``aot_eager``: Dynamo + AOT autograd) three compiled training steps leave bit-identical parameters to three
    eager ones.  (The default Inductor backend would emit Triton kernels for the traced pieces; that is not this package's path to a
    launch-free step: HIP-graph capture, above, is.)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import graph_step as G

    dev = torch.device("cuda", 0)
    batch = G.make_batch(64, dev, shuffled=shuffled)
    torch._dynamo.reset()
    task_c, opt_c = G.make(dev, 512)
    compiled = torch.compile(task_c.training_step, backend=backend, fullgraph=fullgraph)
    for _ in range(3):
        opt_c.zero_grad(set_to_none=False)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = compiled(batch, 0)
        loss.backward()
        opt_c.step()
    task_e, opt_e = G.make(dev, 512)
    for _ in range(3):
        G.step(task_e, opt_e, batch)
    torch.cuda.synchronize()
    assert torch.isfinite(loss.detach()).all()
    for pc, pe in zip(task_c.parameters(), task_e.parameters()):
        assert torch.equal(pc.detach(), pe.detach())
    torch._dynamo.reset()



def test_eager_steps_between_replays_cannot_free_what_the_graph_reads_and_capture_refuses_moving_gradients():
    """a captured AdamW launch bakes in the addresses of the optimizer's device tables.  An eager step with
    other gradient addresses (``zero_grad(set_to_none=True)``: a ragged last batch run eagerly) replaces the optimizer's current
    tables; the ones the graph points at must stay alive, and a replay afterwards must still be the step it captured.  Capturing
    with gradients that are not the last eager step's is refused with a clear error instead of allocating pinned memory inside
    the capture."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import graph_step as G

    dev = torch.device("cuda", 0)
    batch = G.make_batch(64, dev)
    task_e, opt_e = G.make(dev, 512, own_adamw=True)
    task_g, opt_g = G.make(dev, 512, own_adamw=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            G.step(task_g, opt_g, batch)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        G.step(task_g, opt_g, batch)
    assert opt_g._captured, "tables used under capture are pinned to the optimizer"
    held = {id(o) for o in opt_g._captured}
    graph.replay()                                   # step 4
    # step 5 eagerly on fresh gradient tensors: the optimizer's current gradient table is replaced ...
    static_grads = [p.grad for p in task_g.parameters()]   # the buffers the graph's backward writes: the caller keeps those alive
    opt_g.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = task_g.training_step(batch, 0)
    loss.backward()
    moved = [p.grad for p in task_g.parameters()]
    opt_g.step()
    assert all(id(t) not in held for t in opt_g._grad_tables.values()), "an eager step on new gradients builds its own table"
    assert {id(o) for o in opt_g._captured} == held
    del moved
    opt_g.zero_grad(set_to_none=True)                 # the eager step's gradients are gone; the graph owns its own
    torch.cuda.synchronize()
    graph.replay()                                   # step 6, through the tables captured before the eager step
    torch.cuda.synchronize()
    for _ in range(6):
        G.step(task_e, opt_e, batch)
    torch.cuda.synchronize()
    for pe, pg in zip(task_e.parameters(), task_g.parameters()):
        assert torch.equal(pe.detach(), pg.detach())
    assert len(static_grads) == len(list(task_g.parameters()))
    # refusal: pretend to capture with gradients at new addresses (no real capture is opened, so nothing is left half-captured)
    opt_g.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        task_g.training_step(batch, 0).backward()
    real = torch.cuda.is_current_stream_capturing
    torch.cuda.is_current_stream_capturing = lambda: True
    try:
        with pytest.raises(RuntimeError, match="static gradient buffers"):
            opt_g.step()
    finally:
        torch.cuda.is_current_stream_capturing = real


@pytest.mark.parametrize("true_ema", [False, True])
def test_ijepa_vit_s_step_with_ema_is_captured_and_replays_bit_identically(true_ema):
    """the I-JEPA step as HIP graphs (``mmlearn_amd.graph.CapturedIJEPAStep``).  The masks are drawn one step
    ahead on the host -- the generator's own call sequence on the global RNG (mmlearn/datasets/processors/masking.py:384-387) --
    staged in pinned memory and uploaded on a side stream; the captured region (teacher, context encoder, predictor, fused target +
    loss, backward, AdamW, EMA update) holds no host RNG and no pageable copy.  ViT-S/16 + 6 x 384 predictor, seven steps on seven
    batches: three eager warm-up calls, then four replays (one graph per mask geometry); student AND teacher end bit-identical to
    seven plain eager steps that sample their masks inside the step, and the EMA schedule (``decay``, ``num_updates``) agrees.
    ``true_ema``: the decay word of the annealed schedule is what the replayed update reads."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_ijepa_step as T
    from mmlearn_amd.graph import CapturedIJEPAStep

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(17)
    images = [torch.rand(32, 3, 224, 224, generator=g).to(dev) for _ in range(7)]

    def make():
        task = T.build("vits", True, dev, capturable=True, true_ema=true_ema)
        opt = task.configure_optimizers()
        return task, (opt["optimizer"] if isinstance(opt, dict) else opt)

    # eager: masks sampled inside every step
    task_e, opt_e = make()
    torch.manual_seed(5)
    for k in range(7):
        opt_e.zero_grad(set_to_none=False)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss_e = task_e.training_step({"rgb": images[k]}, 0)
        loss_e.backward()
        opt_e.step()
        task_e.on_before_zero_grad(opt_e)
    # captured: same host RNG seed, masks staged ahead
    task_g, opt_g = make()
    runner = CapturedIJEPAStep(task_g, opt_g, warmup=3)
    torch.manual_seed(5)
    for k in range(7):
        loss_g = runner({"rgb": images[k]})
    torch.cuda.synchronize()
    assert runner.replays == 4 and 1 <= len(runner.graphs) <= 4
    assert torch.isfinite(loss_g.detach()).all() and float(loss_g.detach()) == float(loss_e.detach())
    for (n, pe), (_, pg) in zip(task_e.named_parameters(), task_g.named_parameters()):
        assert torch.equal(pe.detach(), pg.detach()), n
    for (n, te), (_, tg) in zip(task_e.target_encoder.model.state_dict().items(), task_g.target_encoder.model.state_dict().items()):
        assert torch.equal(te, tg), n
    assert task_e.target_encoder.num_updates == task_g.target_encoder.num_updates == 7
    assert task_e.target_encoder.decay == task_g.target_encoder.decay
    if true_ema:   # the teacher is an average, not a copy, and the word holds the annealed value the next replay will read
        assert not torch.equal(task_g.target_encoder.model.blocks[0].attn.qkv.weight, task_g.encoder.blocks[0].attn.qkv.weight)
        assert abs(float(task_g.target_encoder._decay_word) - task_g.target_encoder.decay) < 1e-7
    # a step that has not staged its masks cannot be captured: clear error, nothing half-captured
    real = torch.cuda.is_current_stream_capturing
    task_g._mask_stage.pending = task_g._mask_stage.ready = None
    torch.cuda.is_current_stream_capturing = lambda: True
    try:
        with pytest.raises(RuntimeError, match="stage_masks"):
            task_g._masks_for_step(32, dev)
    finally:
        torch.cuda.is_current_stream_capturing = real


def test_captured_ijepa_step_with_stock_blocks_and_torch_adamw():
    """The runner does not depend on this package's encoder kernels or optimizer: stock timm-style blocks (SDPA attention, ATen LayerNorm),
    `torch.optim.AdamW(capturable=True)`, the I-JEPA path ops (index gathers, predictor assembly, fused target + loss, EMA) on HIP.
    Library attention / GEMM kernels may reorder sums between an eager launch and a replay, so the comparison is to 1e-4 of the
    largest parameter, not bitwise."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_ijepa_step as T
    from mmlearn_amd.graph import CapturedIJEPAStep

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(23)
    images = [torch.rand(8, 3, 224, 224, generator=g).to(dev) for _ in range(6)]

    def make():
        task = T.build(True, False, dev, capturable=True)   # the two-block toy ViT, stock modules, torch.optim.AdamW
        opt = task.configure_optimizers()
        return task, (opt["optimizer"] if isinstance(opt, dict) else opt)

    task_e, opt_e = make()
    assert type(opt_e).__module__.startswith("torch.optim")
    torch.manual_seed(9)
    for k in range(6):
        opt_e.zero_grad(set_to_none=False)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss_e = task_e.training_step({"rgb": images[k]}, 0)
        loss_e.backward()
        opt_e.step()
        task_e.on_before_zero_grad(opt_e)
    task_g, opt_g = make()
    runner = CapturedIJEPAStep(task_g, opt_g, warmup=3)
    torch.manual_seed(9)
    for k in range(6):
        loss_g = runner({"rgb": images[k]})
    torch.cuda.synchronize()
    assert runner.replays == 3
    assert abs(float(loss_g.detach()) - float(loss_e.detach())) <= 1e-3 * max(1.0, abs(float(loss_e.detach())))
    scale = max(p.detach().abs().max().item() for p in task_e.parameters())
    for (n, pe), (_, pg) in zip(task_e.named_parameters(), task_g.named_parameters()):
        assert (pe.detach() - pg.detach()).abs().max().item() <= 1e-4 * scale, n
    assert task_e.target_encoder.num_updates == task_g.target_encoder.num_updates == 6


def test_accelerated_hf_towers_compile_with_fullgraph_as_one_operator_each():
    """An encoder patched by ``accelerate_encoder`` (fused QKV attention, residual add + LayerNorm, MLP GEMM epilogues ... dozens of
    autograd Functions over ctypes kernels) is ONE operator to the tracer (``mmlearn_amd::tower_fwd`` / ``tower_bwd``): the step of the
    small ViT + BERT task -- both towers accelerated, the text tower with the tokenizer's padding mask -- compiles with
    ``fullgraph=True`` on ``aot_eager`` and three compiled training steps leave the parameters three eager steps leave
    (the reference plumbs whole-task compilation, mmlearn/cli/run.py:139)."""
    import bench
    from mmlearn_amd import ContrastiveLoss
    from mmlearn_amd.compiled import is_opaque_tower

    dev = torch.device("cuda", 0)
    batch = bench.synthetic_batch(16, 0, dev, padded=True)
    torch._dynamo.reset()

    def make():
        task = bench.build_task(ContrastiveLoss(), small=True, fused=True).to(dev)
        task.concurrent_encoders = False
        for m in task.modules():   # BERT's dropout draws random numbers: off, so that the two runs can be compared bit for bit
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        for m in task.modules():
            cfg = getattr(m, "config", None)
            if cfg is not None and hasattr(cfg, "attention_probs_dropout_prob"):
                cfg.attention_probs_dropout_prob = 0.0
        return task, task.configure_optimizers()

    task_c, opt_c = make()
    assert all(is_opaque_tower(m) for m in task_c.encoders.values())
    step_c = torch.compile(task_c.training_step, backend="aot_eager", fullgraph=True)
    for _ in range(3):
        opt_c.zero_grad(set_to_none=False)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss_c = step_c(batch, 0)
        loss_c.backward()
        opt_c.step()
    task_e, opt_e = make()
    for _ in range(3):
        opt_e.zero_grad(set_to_none=False)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss_e = task_e.training_step(batch, 0)
        loss_e.backward()
        opt_e.step()
    torch.cuda.synchronize()
    assert torch.isfinite(loss_c.detach()).all() and float(loss_c.detach()) == float(loss_e.detach())
    for (n, pc), pe in zip(task_c.named_parameters(), task_e.parameters()):
        assert torch.equal(pc.detach(), pe.detach()), n
    torch._dynamo.reset()
