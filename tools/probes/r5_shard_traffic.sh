# one rank's share of the row-sharded loss: device time and HBM-side traffic with the gradient GEMM's unit map (round 5) and without
# (MMK_GRAD_UNIT_MAP=0, debug-switch build)
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5/shard
mkdir -p $O
export MMK_LIB_VARIANT=_dbg
for MAP in 1 0; do
  export MMK_GRAD_UNIT_MAP=$MAP
  python3 tools/bench_loss_shard.py --iters 50 > $O/bench_map$MAP.json 2>/dev/null
  for COLS in 8192 4096 2048; do
    RK=$([ $COLS = 2048 ] && echo 1 || echo 3)
    rocprofv3 --pmc FETCH_SIZE -d $O/f_${MAP}_$COLS -o f --output-format csv -- python3 tools/bench_loss_shard.py --cols $COLS --rank $RK --iters 10 > /dev/null 2>&1
    rocprofv3 --pmc WRITE_SIZE -d $O/w_${MAP}_$COLS -o w --output-format csv -- python3 tools/bench_loss_shard.py --cols $COLS --rank $RK --iters 10 > /dev/null 2>&1
  done
  python3 tools/pmc_traffic_shard.py 2048:$(find $O/f_${MAP}_2048 -name "*counter_collection.csv"):$(find $O/w_${MAP}_2048 -name "*counter_collection.csv") 4096:$(find $O/f_${MAP}_4096 -name "*counter_collection.csv"):$(find $O/w_${MAP}_4096 -name "*counter_collection.csv") 8192:$(find $O/f_${MAP}_8192 -name "*counter_collection.csv"):$(find $O/w_${MAP}_8192 -name "*counter_collection.csv") > $O/pmc_traffic_shard_map$MAP.json
  cat $O/bench_map$MAP.json
  python3 -c "
import json; d=json.load(open('$O/pmc_traffic_shard_map$MAP.json'))
for c in ('cols2048','cols4096','cols8192'): print(c, d[c]['total_hbm_bytes'], {k:v['hbm_bytes_per_launch'] for k,v in d[c]['per_kernel'].items()})"
done
find $O -name "*counter_collection.csv" -size +20M -delete
