#!/bin/bash
# counters of the Tacotron decoder-loop kernels (eager launches): HBM fetch per launch and wave-state mix
export TMPDIR=/tmp
mkdir -p gpurun_out
for grp in "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | tr ' ' '+' | cut -c1-40)
  rm -rf gpurun_out/pmc_taco_tmp
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmc_taco_tmp -o p -- python tools/taco_run.py 2 > gpurun_out/pmc_taco.log 2>&1
  echo "$tag rc=$?"
  python tools/pmc_summary.py gpurun_out/pmc_taco_tmp gpurun_out/pmc_taco_$tag.json | grep -E "rnn_rowtile_kernel<2|lsa_fast|rnn_rowtile_kernel<1, 1, 8|rnn_rowtile_kernel<0, 1, 8" | head -6
done
rm -rf gpurun_out/pmc_taco_tmp
