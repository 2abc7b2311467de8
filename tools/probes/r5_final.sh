# final measurement set of round 5 (product library): shard traffic PMC, default bench line
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5/final
mkdir -p $O
for COLS in 8192 4096 2048; do
  RK=$([ $COLS = 2048 ] && echo 1 || echo 3)
  rocprofv3 --pmc FETCH_SIZE -d $O/f_$COLS -o f --output-format csv -- python3 tools/bench_loss_shard.py --cols $COLS --rank $RK --iters 10 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $O/w_$COLS -o w --output-format csv -- python3 tools/bench_loss_shard.py --cols $COLS --rank $RK --iters 10 > /dev/null 2>&1
done
python3 tools/pmc_traffic_shard.py 2048:$(find $O/f_2048 -name "*counter_collection.csv"):$(find $O/w_2048 -name "*counter_collection.csv") 4096:$(find $O/f_4096 -name "*counter_collection.csv"):$(find $O/w_4096 -name "*counter_collection.csv") 8192:$(find $O/f_8192 -name "*counter_collection.csv"):$(find $O/w_8192 -name "*counter_collection.csv") > $O/r05_pmc_traffic_shard.json
find $O -name "*counter_collection.csv" -delete
python3 tools/bench_loss_shard.py --iters 50 > $O/bench_loss_shard.json 2>/dev/null
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -1 $O/bench_default.json | cut -c1-300
