"""Per-kernel medians out of a rocprofv3 --kernel-trace CSV, split by launch grid (so shapes are told apart).
    python tools/trace_summary.py gpurun_out/prof_loss/loss_kernel_trace.csv [substring]"""
import csv, sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else "mmk"
d = defaultdict(list)
for r in rows:
    nm = r["Kernel_Name"]
    if flt not in nm:
        continue
    short = nm.split("(")[0][-60:]
    d[(short, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    v2 = sorted(v)
    print(f"{k[0][:58]:58s} grid {k[1]:>8s},{k[2]:>3s},{k[3]:>2s} n={len(v):4d} med={v2[len(v2) // 2]:9.2f} min={v2[0]:9.2f} us")
