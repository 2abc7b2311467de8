# round-6 profile set (run on the GPU box from the repo root): kernel stats of the headline step (two streams, one stream), of the
# configs[3] leg, of the padded-text step, and of the sharded loss share
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6/prof
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/step -o step --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-eager-leg --no-extra-legs > $O/step.json 2> $O/step.err
MMK_BENCH_NO_STREAMS=1 rocprofv3 --kernel-trace --stats -d $O/step1 -o step1 --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-eager-leg --no-extra-legs > $O/step1.json 2> $O/step1.err
rocprofv3 --kernel-trace --stats -d $O/tt -o tt --output-format csv -- python3 bench.py --leg three_tower > $O/tt.json 2> $O/tt.err
MMK_BENCH_NO_STREAMS=1 rocprofv3 --kernel-trace --stats -d $O/padded -o padded --output-format csv -- python3 bench.py --leg padded_text > $O/padded.json 2> $O/padded.err
rocprofv3 --kernel-trace --stats -d $O/shard -o shard --output-format csv -- python3 bench.py --leg loss_shard > $O/shard.json 2> $O/shard.err
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
ls -la $O $O/*
