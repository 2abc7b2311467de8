"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (sums over dispatches, ratios to SQ_WAVE_CYCLES)."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    n = r["Kernel_Name"][:60]
    if pat in n:
        agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
for n, d in agg.items():
    print(n)
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1
    for k, v in sorted(d.items()):
        print(f"    {k:28s} {v:12.4e}  {v / wc:6.3f} x WAVE_CYCLES")
