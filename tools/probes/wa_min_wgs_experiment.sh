set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5/wa
export MMK_LIB_VARIANT=_dbg
for MW in ${MWS:-2048 8192 65536}; do
  export MMK_WIN_MIN_WGS=$MW
  rocprofv3 --kernel-trace --stats -d gpurun_out/r5/wa/kt_$MW -o kt --output-format csv -- python3 tools/prof_window_attn.py > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE -d gpurun_out/r5/wa/f_$MW -o f --output-format csv -- python3 tools/prof_window_attn.py > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE -d gpurun_out/r5/wa/w_$MW -o w --output-format csv -- python3 tools/prof_window_attn.py > /dev/null 2>&1
  echo "== min_wgs $MW"
  python3 tools/prof_window_attn.py --summarise $(find gpurun_out/r5/wa/f_$MW -name "*counter_collection.csv") $(find gpurun_out/r5/wa/w_$MW -name "*counter_collection.csv") | grep -E "ratio|hbm_bytes"
  grep -h win_attn $(find gpurun_out/r5/wa/kt_$MW -name "*kernel_stats.csv") | cut -d, -f1-4 | cut -c1-120
done
