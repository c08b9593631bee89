#!/bin/bash
# WaveRNN fast-chain HBM counters.  rocprofv3 counter mode segfaults inside its dispatch interception when the five
# launches of the fast chain run interleaved (the general chain and every fast kernel alone are fine), so each launch
# type is profiled ALONE (MBHIP_DIAG=wf_which = its bit: the other four launches of every step are skipped; same
# grid, same arguments, same weights and tables -- FETCH/WRITE do not depend on the values) on the benchmarked
# configs[1] shape (23 folds x 9600 steps).  fc2+hh2 (bit 8) has the byte counts of fc1+hh1 (bit 4).
export TMPDIR=/tmp MBHIP_NO_GRAPH=1
mkdir -p gpurun_out
for ctr in FETCH_SIZE WRITE_SIZE; do
  for m in 1 2 4 16; do
    rm -rf gpurun_out/pmc2_tmp
    MBHIP_DIAG=wf_which=$m timeout -k 5 150 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/pmc2_tmp -o p -- python tools/wrn_run.py 1000 1 > gpurun_out/pmc2_wavernn_${ctr}_$m.log 2>&1
    echo "$ctr mask $m rc=$?"
    python tools/pmc_summary.py gpurun_out/pmc2_tmp gpurun_out/pmc2_wavernn_${ctr}_$m.json | grep "wf_" | head -3
  done
done
rm -rf gpurun_out/pmc2_tmp
