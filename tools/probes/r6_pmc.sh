# round-6 PMC passes (separate passes per counter group, --pmc only): HBM-side traffic of the loss kernels (one-launch kernel at
# N = 1024, tiled path at N = 8192), of one rank's share at C = 8192, and the SQ wait counters of the windowed-attention kernels
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6/pmc
mkdir -p $O
rocprofv3 --pmc FETCH_SIZE -d $O/lf -o f --output-format csv -- python3 tools/bench_loss.py --n 1024 8192 --iters 10 > $O/loss_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/lw -o w --output-format csv -- python3 tools/bench_loss.py --n 1024 8192 --iters 10 > $O/loss_w.log 2>&1
python3 tools/pmc_traffic.py $(find $O/lf -name "*counter_collection.csv") $(find $O/lw -name "*counter_collection.csv") > $O/pmc_traffic.json
rocprofv3 --pmc FETCH_SIZE -d $O/sf -o f --output-format csv -- python3 tools/bench_loss_shard.py --cols 8192 --rank 3 --iters 10 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/sw -o w --output-format csv -- python3 tools/bench_loss_shard.py --cols 8192 --rank 3 --iters 10 > /dev/null 2>&1
python3 tools/pmc_traffic_shard.py 8192:$(find $O/sf -name "*counter_collection.csv"):$(find $O/sw -name "*counter_collection.csv") > $O/pmc_traffic_shard.json
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --kernel-trace -d $O/wa -o wa --output-format csv -- python3 tools/prof_window_attn.py > $O/wa.log 2>&1
python3 tools/pmc_summary.py $(find $O/wa -name "*counter_collection.csv") win_attn > $O/window_attn_sq.txt
python3 - <<PY
import csv,glob
f=glob.glob("$O/wa/**/*kernel_trace.csv",recursive=True)[0]
for key in ("win_attn_fwd","win_attn_bwd"):
    d=[(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open(f)) if key in r["Kernel_Name"]]
    print(key,"us:",[round(x,1) for x in d], "grid/wg", [ (r["Grid_Size"], r["Workgroup_Size"], r.get("VGPR_Count"), r.get("LDS_Block_Size")) for r in csv.DictReader(open(f)) if key in r["Kernel_Name"]][:1])
PY
cat $O/window_attn_sq.txt
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
python3 -c "
import json; d=json.load(open('$O/pmc_traffic.json')); print({k:d.get(k) for k in ('n1024','n8192')})
d=json.load(open('$O/pmc_traffic_shard.json')); print(d['cols8192']['total_hbm_bytes'], {k:v['hbm_bytes_per_launch'] for k,v in d['cols8192']['per_kernel'].items()})"
