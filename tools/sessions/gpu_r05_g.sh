#!/bin/bash
# per-kernel durations of the Tacotron loop forms (eager launches under rocprofv3 --kernel-trace --stats)
exec < /dev/null
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "" "taco_front=0"; do
  rm -rf gpurun_out/prof_tl
  MBHIP_NO_GRAPH=1 MBHIP_DIAG="$v" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tl -o tl -- python tools/taco_run.py > gpurun_out/r05_prof_tl.log 2>&1
  f=$(find gpurun_out/prof_tl -name '*kernel_stats*' | head -1)
  echo "== MBHIP_DIAG=$v"; cut -d, -f1-4 "$f" | grep -i "taco\|lsa" | cut -c1-120 | head -12
done
rm -rf gpurun_out/prof_tl
