"""Experiment: the N = 8192 one-rank loss path with the backward's mirrored pair run as two independent directions, i.e. both on the
one-kernel backward (csrc/clip_bwd.hip), against the paired form (one tile pass -> G, grad_gemm + transposed-read kernel)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from mmlearn_amd import kernels as K

dev = torch.device("cuda", 0)
out = {}
out["paired"] = bench.loss_n8192_leg(dev)
orig = K._pair_up
K._pair_up = lambda dirs, backward=False: orig(dirs, backward) if not backward else [(d, None) for d in dirs]
out["unpaired_backward"] = bench.loss_n8192_leg(dev)
for k, v in out.items():
    print(k, v["device_us_fwd_bwd"], v["wall_us_fwd_bwd"], v["loss_path_kernel_us"])
