"""A/B: backward of a mirrored pair on one rank as two one-kernel directions (kernels.ONE_KERNEL_PAIRS) against the tied form
(one tile pass -> G, grad_gemm + the transposed-read kernel), over batch sizes beyond the one-launch path."""
import io, json, os, sys, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
import bench_loss
from mmlearn_amd import kernels as K

res = {}
for n in [int(x) for x in (sys.argv[1:] or ["1536", "2048", "3072", "4096", "8192"])]:
    for flag in (True, False, True, False):
        K.ONE_KERNEL_PAIRS = flag
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            bench_loss.run(n, 512, "bf16", 20, path="tiled")
        d = json.loads(buf.getvalue().strip().splitlines()[-1])
        res.setdefault(n, {}).setdefault("one_kernel" if flag else "tied", []).append((d["device_us_total"], d["wall_ms"], d["kernel_us"]))
for n, v in res.items():
    for k, runs in v.items():
        print(n, k, [r[0] for r in runs], [r[1] for r in runs], runs[-1][2])
