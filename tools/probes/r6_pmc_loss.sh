# round-6 re-take of the loss kernels' HBM-side traffic after the one-kernel backward (separate --pmc passes; tools/probes/r6_pmc.sh
# holds the first take and the SQ counters of the windowed attention)
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6/pmc2
mkdir -p $O
rocprofv3 --pmc FETCH_SIZE -d $O/lf -o f --output-format csv -- python3 tools/bench_loss.py --n 1024 8192 --iters 10 > $O/loss_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/lw -o w --output-format csv -- python3 tools/bench_loss.py --n 1024 8192 --iters 10 > $O/loss_w.log 2>&1
python3 tools/pmc_traffic.py $(find $O/lf -name "*counter_collection.csv") $(find $O/lw -name "*counter_collection.csv") > $O/pmc_traffic.json
find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
python3 -c "
import json; d=json.load(open('$O/pmc_traffic.json')); print({k:d.get(k) for k in ('n1024','n8192')})"
