"""GPU: the bench's training step under DistributedDataParallel with every encoder-side fusion switched on.

Two worker processes share cuda:0 and talk over gloo (RCCL refuses two ranks on one device; the loss's device
collectives are staged through the host as in test_dist_gpu).  Checks that the patched HF encoders, the custom autograd
functions (fused QKV attention, add + LayerNorm, bias + activation, HIP weight gradient) and the row-sharded loss run
under DDP, give finite losses, and leave identical (all-reduced) gradients on both ranks.
"""

import os
import sys
import traceback

import numpy as np
import pytest
import torch
import torch.distributed as dist

import mp_util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, streams, q, done):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        for p in (ROOT, os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        dist.init_process_group("gloo", rank=rank, world_size=world)   # the loss's collectives run on device tensors as they are
        import bench
        from mmlearn_amd import ContrastiveLoss

        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        torch.manual_seed(0)  # same initial weights on both ranks
        task = bench.build_task(ContrastiveLoss(static_shapes=True), small=True, fused=True).to(dev)
        task.concurrent_encoders = streams
        if streams:   # one DDP instance per tower, built under the tower's stream (what bench.py does at N > 1)
            task.wrap_towers_in_ddp()
            stepper = bench._Step(task)
        else:
            stepper = torch.nn.parallel.DistributedDataParallel(bench._Step(task), device_ids=[0])
        opt = task.configure_optimizers()
        batch = bench.synthetic_batch(1024, rank, dev)   # 1024 x 17 tokens >= 6k rows: the wgrad path is live
        losses = []
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = stepper(batch)
            loss.backward()
            gsum = torch.stack([p.grad.float().abs().sum() for p in task.parameters() if p.grad is not None]).sum()
            opt.step()
            losses.append((float(loss.detach().float().item()), float(gsum.item())))
        item = (rank, losses, None)
    except Exception:  # pragma: no cover
        item = (rank, None, traceback.format_exc())
    try:
        mp_util.send(q, done, item)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("streams", [False, True])
def test_ddp_step_with_fused_encoders(streams):
    res = mp_util.run(_worker, 2, lambda r, port: (r, 2, port, streams))
    res.sort(key=lambda r: r[0])
    for rank, losses, err in res:
        assert err is None, f"rank {rank}:\n{err}"
    (l0, g0), (l1, g1) = res[0][1][0], res[1][1][0]
    assert all(map(lambda v: v == v and abs(v) < 1e6, (l0, l1, g0, g1)))
    assert abs(g0 - g1) <= 1e-3 * max(g0, 1e-6)          # DDP left the same averaged gradients on both ranks
    assert res[0][1][1][0] == res[0][1][1][0]            # second step finite


def _nccl_world1_worker(port, q, done):
    """One rank over RCCL (world_size 1 still goes through DDP's reducer: gradient hooks, bucket copies, the collective's
    stream hand-off) -- the real process-group type of the bench, which two ranks on one device cannot use."""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        if ROOT not in sys.path:
            sys.path.insert(0, ROOT)
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        import bench
        from mmlearn_amd import ContrastiveLoss
        from mmlearn_amd import kernels as K

        dev = torch.device("cuda", 0)
        out = []
        # (tower streams, gather path forced on the 1-rank group, static_shapes): the last two run the packed all-gathers of
        # the N > 1 loss over real RCCL -- prefetched from the towers' streams, and in forward() behind the size header.
        # All four on the tiled loss kernels (what every gathered problem runs), so that only f32 atomics may reorder between
        # them; the fifth repeats the second on the one-launch loss (other summation orders: compared at 2e-3).
        for streams, gather, static, fused in ((False, False, True, False), (True, False, True, False), (True, True, True, False),
                                               (True, True, False, False), (True, False, True, True)):
            loss_fn = ContrastiveLoss(static_shapes=static)
            with K.seams(FUSED_LOSS=fused), loss_fn.forcing_gather(gather):
                task = bench.build_task(loss_fn, small=True, fused=True).to(dev)
                task.eval()   # dropout off
                task.concurrent_encoders = streams
                if streams:   # per-tower DDP instances with many small buckets, bucket-view gradients as in bench.py
                    task.wrap_towers_in_ddp(bucket_cap_mb=1)
                    stepper = bench._Step(task)
                else:
                    stepper = torch.nn.parallel.DistributedDataParallel(bench._Step(task), device_ids=[0], bucket_cap_mb=1,
                                                                        gradient_as_bucket_view=True)
                batch = bench.synthetic_batch(1024, 0, dev)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    loss = stepper(batch)
                loss.backward()
                torch.cuda.synchronize()
                g = torch.cat([p.grad.detach().float().flatten() for p in task.parameters() if p.grad is not None])
                out.append((float(loss.detach().float().item()), g.cpu().numpy(), loss_fn.prefetched_gathers_used, loss_fn.prefetched_matches_used))
        item = (out, None)
    except Exception:  # pragma: no cover
        item = (None, traceback.format_exc())
    try:
        mp_util.send(q, done, item)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_ddp_over_rccl_gives_the_same_gradients_with_tower_streams():
    (out, err), = mp_util.run(_nccl_world1_worker, 1, lambda r, port: (port,))
    assert err is None, err
    assert [o[2] for o in out] == [0, 0, 1, 0, 0]   # only the static-shapes gather variant takes the prefetched collectives
    assert [o[3] for o in out] == [1, 1, 1, 0, 1]   # matcher ahead of the encoders; not behind a size header
    out = [o[:2] for o in out]
    (l0, g0) = out[0]
    scale = float(np.abs(g0).max())
    assert scale > 0
    # same weights, same batch.  The tiled-loss variants are not bitwise reproducible (f32 atomics in the embedding backward; the
    # library's stream-K GEMM selections): six repetitions gave 0.8e-6 ... 2.2e-6 of the largest gradient, a different figure
    # every time, and the former 1e-5 bound failed once in eight runs of the suite.  1e-4 now -- a schedule bug (a tower reading
    # a buffer too early) is orders of magnitude above that.  One-launch variant: bf16 G tiles, 1.5e-4 measured, bound 2e-3.
    for k, (l, g) in enumerate(out[1:]):
        tol = 2e-3 if k == 3 else 1e-4
        assert abs(l0 - l) <= tol * max(1.0, abs(l0)), (k, l0, l)
        assert float(np.abs(g0 - g).max()) <= tol * scale, (k, float(np.abs(g0 - g).max()) / scale)


def _three_tower_worker(port, q, done):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        if ROOT not in sys.path:
            sys.path.insert(0, ROOT)
        from functools import partial

        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        from mmlearn_amd import ContrastiveLoss
        from mmlearn_amd.tasks import ContrastivePretraining, LossPairSpec, ModuleKeySpec

        dev = torch.device("cuda", 0)
        D, B = 64, 256

        class Enc(torch.nn.Module):
            def __init__(self, key):
                super().__init__()
                self.key = key
                self.net = torch.nn.Sequential(torch.nn.Linear(96, 128), torch.nn.GELU(), torch.nn.Linear(128, D))

            def forward(self, inputs):
                return (self.net(inputs[self.key]),)

        g = torch.Generator().manual_seed(8)
        batch = {m: torch.randn(B, 96, generator=g).to(dev) for m in ("rgb", "text", "audio")}
        ids = torch.stack([torch.zeros(B, dtype=torch.long), torch.arange(B)], 1).to(dev)
        batch["example_ids"] = {m: ids.clone() for m in ("rgb", "text", "audio")}
        out = []
        for ddp in (False, True):
            torch.manual_seed(3)
            task = ContrastivePretraining(
                encoders={"rgb": Enc("rgb"), "text": Enc("text"), "audio": Enc("audio")},
                heads={"shared": {"proj": torch.nn.Linear(D, D)}},
                modality_module_mapping={m: ModuleKeySpec(encoder_key=m, head_key="shared") for m in ("rgb", "text", "audio")},
                loss=ContrastiveLoss(static_shapes=True), optimizer=partial(torch.optim.SGD, lr=0.1),
                modality_loss_pairs=[LossPairSpec(("rgb", "text")), LossPairSpec(("rgb", "audio"), 0.5)],
                compute_validation_loss=False, compute_test_loss=False).to(dev)
            task.concurrent_encoders = True
            if ddp:
                task.wrap_towers_in_ddp(bucket_cap_mb=1)
                # the wrappers live outside the module tree (checkpoint keys stay the reference's): one per encoder, and the
                # shared head is wrapped once (by the first tower); the other towers call the same module unwrapped
                assert all(type(v).__name__ != "DistributedDataParallel" for v in list(task.encoders.values()) + list(task.heads.values()))
                assert sorted(k for k in task._tower_ddp if k[0] == "encoders") == [("encoders", m) for m in ("audio", "rgb", "text")]
                assert sum(k[0] == "heads" for k in task._tower_ddp) == 1
                assert not any(".module." in k for k in task.state_dict())
            rec = []
            opt = task.configure_optimizers()
            for _ in range(2):
                opt.zero_grad(set_to_none=True)
                loss = task.training_step(batch, 0)
                loss.backward()
                torch.cuda.synchronize()
                names = sorted(n.replace("module.", "") for n, p in task.named_parameters() if p.grad is not None)
                grads = {n.replace("module.", ""): p.grad.detach().clone() for n, p in task.named_parameters() if p.grad is not None}
                rec.append((float(loss), names, torch.cat([grads[n].flatten() for n in names]).float().cpu().numpy()))
                opt.step()
            out.append(rec)
        item = (out, None)
    except Exception:  # pragma: no cover
        item = (None, traceback.format_exc())
    try:
        mp_util.send(q, done, item)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_per_tower_ddp_with_three_towers_and_a_shared_head():
    (out, err), = mp_util.run(_three_tower_worker, 1, lambda r, port: (port,))
    assert err is None, err
    plain, wrapped = out
    for (l0, n0, g0), (l1, n1, g1) in zip(plain, wrapped):
        assert n0 == n1 and "log_logit_scale" in n0
        assert abs(l0 - l1) <= 1e-6 * max(1.0, abs(l0))
        # Measured (round 4, MMK_TEST_SPREAD_OUT, six runs x two steps on one MI355X): the plain and the per-tower-DDP step of
        # THIS configuration (MLP towers: no embedding-backward atomics, no library stream-K GEMMs) agree bit for bit, 0.0 in all
        # twelve comparisons.  The bound stays at the two-tower test's 1e-4 allowance for towers that do have such kernels: what
        # this test guards against -- a tower reading a bucket too early, a missing shared-head all-reduce -- is orders above it.
        rel = float(np.abs(g0 - g1).max()) / float(np.abs(g0).max())
        if os.environ.get("MMK_TEST_SPREAD_OUT"):   # measurement runs: append the observed difference (tools: see the comment above)
            with open(os.environ["MMK_TEST_SPREAD_OUT"], "a") as f:
                f.write(f"{rel:.3e}\n")
        assert rel <= 1e-4
