#!/bin/bash
# HBM-traffic counters of the WaveRNN loop kernels (rocprofv3 counter mode crashes on hipGraph replays,
# so the loop runs eagerly: same kernels, same arguments).
export TMPDIR=/tmp MBHIP_NO_GRAPH=1
mkdir -p gpurun_out
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$ctr
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/pmc_$ctr -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-hifigan --no-tacotron --no-ppg2mel --no-wavernn-batch --frames 200 > gpurun_out/pmc_$ctr.log 2>&1
  echo "$ctr rc=$?"
  python tools/pmc_summary.py gpurun_out/pmc_$ctr gpurun_out/pmc_${ctr}_summary.json | head -8
  rm -rf gpurun_out/pmc_$ctr
done
