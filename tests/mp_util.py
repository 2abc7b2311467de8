"""Process plumbing shared by the multi-process tests (GPU and gloo-on-CPU).

Three rules, each the cure for a failure seen on a driver box:

* results cross the process boundary as plain data (numpy arrays, python scalars, strings) -- never a `torch.Tensor`: a tensor is
  pickled as a file descriptor served by the SENDER's resource-sharer socket, so a parent that unpickles after the child has
  exited gets `FileNotFoundError` (`send()` refuses tensors outright);
* a child stays alive until the parent has read every result (`done` event), so nothing the queue's feeder thread still holds can
  be torn down under the parent;
* rendezvous ports come from the OS (`bind(0)`), not from a pid-derived constant.
"""

import queue as _queue
import socket
import time
import traceback

import numpy as np
import torch
import torch.multiprocessing as mp


_NOT_PLAIN = "__mp_util_not_plain__"


def free_port():
    """A TCP port nobody is listening on right now (asked from the OS)."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def assert_plain(obj, path="result"):
    """Raise if `obj` holds a torch.Tensor anywhere (dicts / lists / tuples walked)."""
    if isinstance(obj, torch.Tensor):
        raise TypeError(f"{path} is a torch.Tensor: send .numpy() / float() across processes (see tests/mp_util.py)")
    if isinstance(obj, dict):
        for k, v in obj.items():
            assert_plain(k, f"{path} key")
            assert_plain(v, f"{path}[{k!r}]")
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            assert_plain(v, f"{path}[{i}]")
    elif not isinstance(obj, (np.ndarray, np.generic, int, float, bool, str, bytes, type(None))):
        raise TypeError(f"{path}: unexpected {type(obj).__name__} in a cross-process result")


def send(q, done, item, wait_s=120.0):
    """Child side: put `item` (plain data only) and hold the process until the parent says it has everything."""
    try:
        assert_plain(item)
    except TypeError:
        item = (_NOT_PLAIN, traceback.format_exc())
    q.put(item)
    done.wait(wait_s)


def run(target, n_procs, args_of_rank, n_results=None, timeout=600.0):
    """Start `n_procs` spawn children `target(*args_of_rank(rank, port), q, done)`, collect `n_results` queue items (default one per
    child), release the children and join them.  A child that dies without reporting fails the test at once instead of
    after the queue timeout."""
    ctx = mp.get_context("spawn")
    q, done = ctx.Queue(), ctx.Event()
    port = free_port()
    procs = [ctx.Process(target=target, args=tuple(args_of_rank(r, port)) + (q, done)) for r in range(n_procs)]
    for p in procs:
        p.start()
    want = n_procs if n_results is None else n_results
    items, t_end = [], time.time() + timeout
    try:
        while len(items) < want:
            try:
                item = q.get(timeout=2.0)
                if isinstance(item, tuple) and len(item) == 2 and item[0] == _NOT_PLAIN:
                    raise AssertionError(item[1])
                items.append(item)
                continue
            except _queue.Empty:
                pass
            dead = [p for p in procs if p.exitcode not in (None, 0)]
            if dead:
                raise AssertionError(f"worker pid {dead[0].pid} exited with code {dead[0].exitcode} before reporting "
                                     f"({len(items)}/{want} results in)")
            if time.time() > t_end:
                raise AssertionError(f"timed out after {timeout:.0f} s with {len(items)}/{want} results")
    finally:
        done.set()
        for p in procs:
            p.join(timeout=60)
        for p in procs:
            if p.is_alive():   # exact child we started, never a pattern
                p.kill()
                p.join(timeout=10)
    return items
