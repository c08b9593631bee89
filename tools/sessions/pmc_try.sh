#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for variant in nograph graph; do
  for ctr in FETCH_SIZE; do
    rm -rf gpurun_out/pmc_$variant
    if [ $variant = nograph ]; then export MBHIP_NO_GRAPH=1; else unset MBHIP_NO_GRAPH; fi
    timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/pmc_$variant -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-hifigan --no-tacotron --frames 60 > gpurun_out/pmc_$variant.log 2>&1
    echo "$variant $ctr rc=$?"
    find gpurun_out/pmc_$variant -name '*.csv' | head -3
    python tools/pmc_summary.py gpurun_out/pmc_$variant gpurun_out/pmc_${variant}_summary.json 2>&1 | head -8
    rm -rf gpurun_out/pmc_$variant
  done
done
