"""One tower of the configs[3] leg alone: forward + backward, same modules and batch as ``bench.py --leg three_tower``.
For `rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/prof_tower.py --tower audio` (where do HTSAT's
145 ms go) and for A/B of what ``accelerate_encoder`` swaps in the tower (``--stock``).

    python tools/prof_tower.py --tower audio [--stock] [--batch 256] [--passes 3]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tower", default="audio", choices=["rgb", "text", "audio"])
    ap.add_argument("--stock", action="store_true")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--small", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    tower = {"rgb": bench._PooledVision, "text": bench._PooledText, "audio": bench._PooledAudio}[a.tower](a.small)
    if not a.stock:
        bench.accelerate_tower(tower, a.tower)
    tower = tower.to(dev)
    batch = bench._three_tower_batch(a.batch, dev)

    def one():
        tower.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            e = tower(batch)[0]
        e.float().sum().backward()

    one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.passes):
        one()
    torch.cuda.synchronize()
    print({"tower": a.tower, "stock": a.stock, "batch": a.batch, "fwd_bwd_ms": round((time.perf_counter() - t0) / a.passes * 1e3, 2)})


if __name__ == "__main__":
    main()
