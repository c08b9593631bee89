#!/bin/bash
# isolate the costs of the fused pair kernel: per-kernel rocprof stats under each MBHIP_DIAG=pair_dbg bit
export TMPDIR=/tmp
mkdir -p gpurun_out
for dbg in ${DBGS:-0 1 2 4 7}; do
  rm -rf gpurun_out/prof_dbg
  MBHIP_DIAG=pair_dbg=$dbg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dbg -o gan -- python tools/gan_run.py hifigan f16 32 200 5 > gpurun_out/prof_dbg.log 2>&1
  grep batch gpurun_out/prof_dbg.log | sed "s/^/dbg=$dbg /"
  f=$(find gpurun_out/prof_dbg -name '*kernel_stats*' | head -1); grep resblock_pair "$f" | python3 -c "
import sys,csv
for r in csv.reader(sys.stdin): print('dbg=$dbg', r[0][22:50], 'calls', r[1], 'avg_us', round(float(r[3])/1000,1), 'min', round(float(r[5])/1000,1), 'max', round(float(r[6])/1000,1))"
done
rm -rf gpurun_out/prof_dbg
