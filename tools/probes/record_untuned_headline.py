"""Library GEMM ops of the headline step (configs[1], batch 1024) without an entry in the shipped selections file (record-untuned mode)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.cuda.tunable as tunable
from mmlearn_amd import ContrastiveLoss, tuned
import bench

dev = torch.device("cuda", 0)
os.environ["PYTORCH_TUNABLEOP_UNTUNED_FILENAME"] = os.path.join(ROOT, "gpurun_out", "r5", "untuned_headline.csv")
tunable.enable(True)
tunable.tuning_enable(False)
tunable.read_file(tuned.DEFAULT_FILE)
tunable.record_untuned_enable(True)
task = bench.build_task(ContrastiveLoss(static_shapes=True), False, fused=True).to(dev)
opt = task.configure_optimizers()
batch = bench.synthetic_batch(1024, 0, dev)
for _ in range(2):
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = task.training_step(batch, 0)
    loss.backward()
    opt.step()
torch.cuda.synchronize()
print(float(loss.detach()))
