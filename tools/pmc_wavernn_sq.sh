#!/bin/bash
# SQ counters of the resident WaveRNN kernel (one launch of tools/wrn_run.py 1000 1 = BASELINE configs[1]): how its waves spend their cycles.
# -> gpurun_out/pmc_wavernn_sq_{a,b}.json (tools/pmc_summary.py); summarised into profiles/r04_wavernn_pipe16_sq_counters.json by hand
exec < /dev/null
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU"; do
  tag=$(echo a b | cut -d' ' -f$((i+1))); i=$((i+1))
  rm -rf gpurun_out/pmc6
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc6 -o p -- python tools/wrn_run.py 1000 1 > gpurun_out/pmc_wavernn_sq_$tag.log 2>&1
  echo "set $tag rc=$?"
  timeout 60 python tools/pmc_summary.py gpurun_out/pmc6 gpurun_out/pmc_wavernn_sq_$tag.json | grep "wf_pipe16" | head -2
done
rm -rf gpurun_out/pmc6
