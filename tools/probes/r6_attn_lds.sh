# LDS counters of the attention kernels at the ViT-B/16 shape (B 1024, H 12, L 197): are they bound by LDS bandwidth / bank conflicts?
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6/attnlds
mkdir -p $O
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES --kernel-trace -d $O/a -o a --output-format csv -- python3 tools/bench_attn.py > $O/a.log 2>&1 || true
python3 tools/pmc_summary.py $(find $O/a -name "*counter_collection.csv") attn_ > $O/lds.txt || true
python3 - <<PY
import csv,glob
f=glob.glob("$O/a/**/*kernel_trace.csv",recursive=True)[0]
import collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "attn_" in r["Kernel_Name"]:
        d[r["Kernel_Name"][:70]].append((float(r["End_Timestamp"])-float(r["Start_Timestamp"]))/1e3)
for k,v in d.items(): print(k, len(v), round(sum(v)/len(v),1), "us")
PY
cat $O/lds.txt
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
