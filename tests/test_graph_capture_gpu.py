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
