"""Which kernels precede the runtime's buffer fills (hipMemsetAsync = torch.zeros / zero_) in a rocprofv3 kernel trace, and how big
the fills are.    python tools/fill_sources.py <kernel_trace.csv>"""
import collections, csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
agg = collections.defaultdict(lambda: [0, 0.0, 0])
for i, r in enumerate(rows):
    if "fillBuffer" not in r["Kernel_Name"]:
        continue
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    prev = next((rows[j]["Kernel_Name"][:70] for j in range(i - 1, max(i - 6, -1), -1) if "fillBuffer" not in rows[j]["Kernel_Name"]), "?")
    nxt = next((rows[j]["Kernel_Name"][:70] for j in range(i + 1, min(i + 6, len(rows))) if "fillBuffer" not in rows[j]["Kernel_Name"]), "?")
    key = (prev, nxt, int(r.get("Grid_Size", 0) or 0))
    agg[key][0] += 1
    agg[key][1] += dur
for (prev, nxt, grid), (n, t, _) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{n:5d} x {t / n:8.1f} us  grid {grid:10d}  after [{prev}]  before [{nxt}]")
