set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6/cblds
mkdir -p $O
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN --kernel-trace -d $O/a -o a --output-format csv -- python3 tools/bench_loss_shard.py --cols 8192 --rank 3 --iters 10 > $O/a.log 2>&1 || true
python3 tools/pmc_summary.py $(find $O/a -name "*counter_collection.csv") clip_bwd_fused > $O/lds.txt || true
cat $O/lds.txt; tail -3 $O/a.log
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
