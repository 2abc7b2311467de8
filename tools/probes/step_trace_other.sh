# kernel trace of the single-stream headline step; prints, for the LAST step of the trace, every kernel outside the big families
# (library / own GEMMs, wgrad, LayerNorm family, attention) with its duration and its predecessor -- where do the "other" ms go?
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5/trace
mkdir -p $O
MMK_BENCH_NO_STREAMS=1 rocprofv3 --kernel-trace -d $O -o t --output-format csv -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-eager-leg --no-extra-legs > $O/b.json 2> $O/b.err
python3 - <<PY
import csv, glob
f = glob.glob("$O/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last adamw launch ends a step; take the kernels between the two last adamw groups
idx = [i for i, n in enumerate(names) if "adamw" in n]
end = idx[-1]
prev = [i for i in idx if i < end - 50][-1]
step = rows[prev + 1:end + 1]
big = ("Cijk", "wgrad_kernel", "mlp_gemm", "layernorm", "attn_fwd_kernel", "attn_bwd5", "attn_bwd_kernel")
tot = 0
out = []
for k, r in enumerate(step):
    n = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if any(b in n for b in big):
        continue
    tot += d
    out.append((k, d, n[:90], step[k - 1]["Kernel_Name"][:40] if k else ""))
print("kernels in step", len(step), "other us", round(tot, 1))
for k, d, n, p in out:
    if d >= 15:
        print(f"{k:5d} {d:8.1f} us  {n}   <- after {p}")
import collections
c = collections.Counter()
t = collections.Counter()
for k, d, n, p in out:
    if d < 15:
        c[n[:60]] += 1; t[n[:60]] += d
print("--- small kernels (< 15 us) by name: count, total us")
for n, v in t.most_common(25):
    print(f"{c[n]:5d} {v:8.1f}  {n}")
PY
rm -rf $O/*/*kernel_trace.csv
