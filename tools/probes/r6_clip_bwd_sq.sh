# SQ counters of the one-kernel sharded backward (product build): where its wave cycles go
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6/cbsq
mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --kernel-trace -d $O/a -o a --output-format csv -- python3 tools/bench_loss_shard.py --cols 8192 --rank 3 --iters 10 > $O/a.log 2>&1
python3 tools/pmc_summary.py $(find $O/a -name "*counter_collection.csv") clip_bwd_fused > $O/sq_a.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM --kernel-trace -d $O/b -o b --output-format csv -- python3 tools/bench_loss_shard.py --cols 8192 --rank 3 --iters 10 > $O/b.log 2>&1 || true
python3 tools/pmc_summary.py $(find $O/b -name "*counter_collection.csv") clip_bwd_fused > $O/sq_b.txt || true
cat $O/sq_a.txt $O/sq_b.txt
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
