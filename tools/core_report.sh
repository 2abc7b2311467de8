#!/bin/bash
# Summarise GPU core dumps (gpucore.*) left in the current directory by a faulting process: faulting waves, their kernels and PCs.
# Usage (on the GPU box, from the directory the process ran in):  bash tools/core_report.sh OUT.txt
out=${1:-core_report.txt}
: > "$out"
for c in gpucore.*; do
  [ -f "$c" ] || continue
  echo "== $c ($(stat -c %s "$c") bytes)" >> "$out"
  timeout -k 5 240 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "info agents" -ex "info threads" -ex "thread apply all bt 3" \
    "$(command -v python3)" -c "$c" > "$out.raw" 2>&1
  grep -n -i "fault\|stopped\|signal\|exception\|violation" "$out.raw" | head -50 >> "$out"
  echo "-- head of the raw log" >> "$out"
  head -c 200000 "$out.raw" >> "$out"
  rm -f "$out.raw" "$c"
done
exit 0
