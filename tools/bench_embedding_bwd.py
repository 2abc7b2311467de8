"""nn.Embedding backward at BERT's bench shape (78,848 token rows x 768): the run-summing atomic scatter against the sorted-run
kernel (csrc/encoder_ops.hip), for uniformly random ids, Zipf-distributed ids and the all-equal ids of the token-type table.
    python tools/bench_embedding_bwd.py [--out gpurun_out/embedding_bwd.json]"""
import argparse, json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import kernels as K

ap = argparse.ArgumentParser(); ap.add_argument("--out", default=None); ap.add_argument("--rounds", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda", 0)
rows, d = 1024 * 77, 768
torch.manual_seed(0)
dout = torch.randn(rows, d, device=dev)
zipf = torch.distributions.Categorical(probs=1.0 / torch.arange(1, 30523, dtype=torch.float64)).sample((rows,)).to(dev)
cases = {"uniform_vocab30522": (torch.randint(0, 30522, (rows,), device=dev), 30522), "zipf_vocab30522": (zipf, 30522),
         "all_equal_vocab2": (torch.zeros(rows, dtype=torch.long, device=dev), 2)}
out = []
for name, (ids, vocab) in cases.items():
    res = {}
    ref = torch.zeros(vocab, d, device=dev, dtype=torch.float64).index_add_(0, ids, dout.double())
    for arm, thr in (("atomic_scatter", 1 << 60), ("sorted_runs", 4096)):
        K.EMBEDDING_SORT_MIN_ROWS = thr
        for _ in range(3): dw = K.embedding_bwd(dout, ids, vocab)
        err = (dw.double() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
        ts = []
        for _ in range(args.rounds):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): K.embedding_bwd(dout, ids, vocab)
            e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) * 100)
        res[arm + "_us"] = round(statistics.median(ts), 1); res[arm + "_max_rel_err"] = err
    row = {"ids": name, "rows": rows, "d": d, **res}
    print(json.dumps(row), flush=True); out.append(row)
if args.out:
    json.dump({"tool": "tools/bench_embedding_bwd.py", "note": "times include the zero-fill of dW and, for sorted_runs, torch.sort of the ids", "rows": out}, open(args.out, "w"), indent=1)
