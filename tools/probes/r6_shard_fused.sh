# round-6, one-kernel sharded backward (csrc/clip_bwd.hip): kernel stats of one rank's share, its HBM-side traffic (separate --pmc
# passes), and the A/B against the two-launch form (debug-switch build, MMK_CLIP_BWD_FUSED=0)
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6/shardf
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/shard -o shard --output-format csv -- python3 bench.py --leg loss_shard > $O/shard.json 2> $O/shard.err
rocprofv3 --pmc FETCH_SIZE -d $O/sf -o f --output-format csv -- python3 tools/bench_loss_shard.py --cols 8192 --rank 3 --iters 10 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/sw -o w --output-format csv -- python3 tools/bench_loss_shard.py --cols 8192 --rank 3 --iters 10 > /dev/null 2>&1
python3 tools/pmc_traffic_shard.py 8192:$(find $O/sf -name "*counter_collection.csv"):$(find $O/sw -name "*counter_collection.csv") > $O/pmc_traffic_shard.json
MMK_LIB_VARIANT=_dbg MMK_CLIP_BWD_FUSED=1 python3 bench.py --leg loss_shard > $O/ab_fused.json 2> /dev/null
MMK_LIB_VARIANT=_dbg MMK_CLIP_BWD_FUSED=0 python3 bench.py --leg loss_shard > $O/ab_two_launch.json 2> /dev/null
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
python3 -c "
import json
d=json.load(open('$O/pmc_traffic_shard.json')); print(d['cols8192']['total_hbm_bytes'], {k:v['hbm_bytes_per_launch'] for k,v in d['cols8192']['per_kernel'].items()})
for n in ('ab_fused','ab_two_launch'):
    j=json.loads(open('$O/'+n+'.json').read().strip().splitlines()[-1]); print(n, j['device_us_per_rank_share'], j['wall_us_per_rank_share'], j['loss_path_kernel_us'])"
ls $O/shard
