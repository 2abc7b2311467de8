"""HBM-side traffic of one rank's share of the row-sharded loss path (tools/bench_loss_shard.py) from rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE do not fit one pass).  Bytes = 2 * FETCH_SIZE + WRITE_SIZE with the counters in KiB (gfx950:
FETCH_SIZE tallies 128-byte requests at 64 bytes; /opt/skills/guides/MI355X_MICROARCH.md, HBM section).

    python tools/pmc_traffic_shard.py COLS:fetch.csv:write.csv [COLS:fetch.csv:write.csv ...] > profiles/r04_pmc_traffic_shard.json

Per column count: bytes per launch of every kernel of the share, their sum (one launch of each per step) and the bytes of the
dominant counted MFMA kernel (sim_stats) -- what bench.py's `roofline.traffic` / `roofline_shard` quote."""
import collections, csv, json, sys

NAMES = {"clip_bwd_fused": "clip_bwd_fused", "clip_fwd_shard": "sim_stats", "Li0ELi1E": "sim_stats", "Li1ELi1E": "sim_grad", "Li2ELi1E": "grad_gemm", "lse_merge": "lse_merge", "pack_tr": "pack_rows",
         "grad_finalize": "grad_finalize", "wgrad_kernel": "wgrad", "wgrad_reduce": "wgrad_reduce"}


def load(path, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        nm = next((v for k, v in NAMES.items() if k in r["Kernel_Name"]), None)
        if nm is None:
            continue
        a = agg[nm]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    return agg


out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/bench_loss_shard.py --rows 1024 --cols C, bf16, D = 512: "
               "one rank's share of the row-sharded loss (both directions); HBM-side bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (KiB counters)"}
for spec in sys.argv[1:]:
    cols, fpath, wpath = spec.split(":")
    fetch, write = load(fpath, "FETCH_SIZE"), load(wpath, "WRITE_SIZE")
    per = {}
    for k, (f, nf) in sorted(fetch.items()):
        w, nw = write.get(k, [0.0, 1])
        per[k] = {"launches_profiled": nf, "fetch_kib": round(f / nf, 1), "write_kib": round(w / max(nw, 1), 1),
                  "hbm_bytes_per_launch": int((2 * f / nf + w / max(nw, 1)) * 1024)}
    R, C, D = 1024, int(cols), 512
    out[f"cols{cols}"] = {"rows": R, "cols": C, "d": D, "per_kernel": per,
                          "total_hbm_bytes": sum(v["hbm_bytes_per_launch"] for v in per.values()),
                          "dominant_hbm_bytes_per_launch": per.get("sim_stats", {}).get("hbm_bytes_per_launch"),
                          "operand_bytes": 2 * (C + R) * D * 2 * 2 + 2 * R * D * 2}
print(json.dumps(out, indent=1))
