"""Random shard shapes through the one-kernel backward against the float64 restatement the parity test uses
(tests/test_clip_gpu.py::test_sharded_backward_one_kernel_vs_float64): ragged row blocks / column tiles, label offsets anywhere,
shards that end with the gathered operand, widths 449..512."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_clip_gpu as T
from mmlearn_amd import _lib, kernels as K

rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
done = 0
while done < n:
    c = rng.choice([1024, 1025, 1100, 1536, 2000, 2048, 3000, 3072, 4096, 5000, 6144])
    r = rng.randint(65, min(c, 2200))
    p0 = rng.choice([0, c - r, rng.randint(0, c - r)])
    d = rng.choice([449, 480, 500, 512, 512])
    col = rng.random() < 0.5
    if not K.backward_recomputes_on_chip(r, c, d, _lib.COMPUTE_BF16, 2):
        continue
    T.test_sharded_backward_one_kernel_vs_float64(r, c, p0, d, col)
    done += 1
    print("ok", r, c, p0, d, col, flush=True)
print("all", done, "shapes agree")
