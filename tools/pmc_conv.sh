#!/bin/bash
# wave-state / LDS counters of the fp32 conv kernel inside one Tacotron generate (CBHG convs), eager loop
export TMPDIR=/tmp MBHIP_NO_GRAPH=1
mkdir -p gpurun_out
for grp in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM"; do
  tag=$(echo $grp | tr ' ' '+' | cut -c1-30)
  rm -rf gpurun_out/pmc_conv_tmp
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmc_conv_tmp -o p -- python tools/taco_run.py 1 > gpurun_out/pmc_conv.log 2>&1
  echo "$tag rc=$?"
  python tools/pmc_summary.py gpurun_out/pmc_conv_tmp gpurun_out/pmc_conv_$tag.json | grep -E "conv1d_mfma" | head -4
done
rm -rf gpurun_out/pmc_conv_tmp
