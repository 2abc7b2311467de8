"""HBM-side traffic per launch of the loss-path kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; they do not
fit one pass) of tools/bench_loss.py.  Bytes = 2 * FETCH_SIZE + WRITE_SIZE with the counters in KiB (gfx950: FETCH_SIZE
tallies 128-byte requests at 64 bytes, /opt/skills/guides/MI355X_MICROARCH.md, HBM section).
    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> > profiles/rNN_pmc_traffic.json"""
import collections, csv, json, sys

NAMES = {"clip_fused_kernel": "clip_fused", "clip_bwd_fused": "clip_bwd_fused", "Li0ELi1E": "sim_stats", "Li1ELi1E": "sim_grad", "Li2ELi1E": "grad_gemm", "lse_merge": "lse_merge", "pack_tr": "pack_rows",
         "grad_finalize": "grad_finalize", "wgrad_kernel": "wgrad (dB = G^T A)", "match_small": "match_ids", "match_kernel": "match_ids", "match_scan": "match_ids"}


def load(path, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        nm = next((v for k, v in NAMES.items() if k in r["Kernel_Name"]), None)
        if nm is None:
            continue
        grid = int(r.get("Grid_Size", 0) or 0)
        a = agg[(nm, grid, r["Kernel_Name"][:48])]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    return agg


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/bench_loss.py --n 1024 8192, bf16, D=512, one pair "
               "(both directions from one tile pass); HBM-side bytes per launch = 2*FETCH_SIZE + WRITE_SIZE, counters in KiB",
       "kernels": []}
for key in sorted(fetch):
    f, nf = fetch[key]
    w, nw = write.get(key, [0.0, 1])
    out["kernels"].append({"kernel": key[0], "grid_threads": key[1], "symbol": key[2], "launches": nf, "fetch_kib_per_launch": round(f / nf, 1),
                           "write_kib_per_launch": round(w / max(nw, 1), 1), "hbm_bytes_per_launch": int((2 * f / nf + w / max(nw, 1)) * 1024)})
# the shapes of the bench: N = 1024 -> the one-launch kernel (256 workgroups x 256 threads), else the similarity-statistics kernel of
# the tiled path (64x64 tiles); N = 8192 -> the similarity-statistics kernel (128x128 tiles)
for k in out["kernels"]:
    if k["kernel"] == "clip_fused" and k["grid_threads"] == 256 * 256:
        out["n1024"] = {"kernel": "clip_fused", "hbm_bytes_per_launch": k["hbm_bytes_per_launch"]}
for tag, sym in (("n1024", "Li64ELi64ELi0E"), ("n8192", "Li128ELi128ELi0E")):
    for k in out["kernels"]:
        if k["kernel"] == "sim_stats" and sym in k["symbol"] and tag not in out:
            out[tag] = {"kernel": "sim_stats", "hbm_bytes_per_launch": k["hbm_bytes_per_launch"]}
# N = 8192 per kernel (the bench quotes the figure of whichever MFMA kernel dominates the leg): the one-kernel backward runs only at
# that size in this tool's run
if "n8192" in out:
    per = {"sim_stats": {"hbm_bytes_per_launch": out["n8192"]["hbm_bytes_per_launch"]}}
    for k in out["kernels"]:
        if k["kernel"] == "clip_bwd_fused":
            per["clip_bwd_fused"] = {"hbm_bytes_per_launch": k["hbm_bytes_per_launch"]}
    out["n8192"]["per_kernel"] = per
print(json.dumps(out, indent=1))
