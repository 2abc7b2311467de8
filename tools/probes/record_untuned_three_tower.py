"""Which library GEMM ops the HIP leg of configs[3] issues that have no entry in the shipped selections file (no tuning: TunableOp's
record-untuned mode).  Strided-batched ops in the list mean the leg must NOT be tuned as a whole (mmlearn_amd/tuned/__init__.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.cuda.tunable as tunable
from mmlearn_amd import tuned
import bench

dev = torch.device("cuda", 0)
os.environ["PYTORCH_TUNABLEOP_UNTUNED_FILENAME"] = os.path.join(ROOT, "gpurun_out", "r5", "untuned_three_tower.csv")
tunable.enable(True)
tunable.tuning_enable(False)
tunable.read_file(tuned.DEFAULT_FILE)
tunable.record_untuned_enable(True)
out = bench.three_tower_leg(256, dev, False, steps=1, warmup=1)
print(out["ms_per_step"])
