#!/bin/bash
# SQ counters of conv1d_split_kernel on ONE conv shape (default: the 128-channel ResBlock conv of HiFi-GAN 32 x 200: T = 5000, k = 7, dil 3).
# usage: bash tools/pmc_conv_split.sh [B Cin Cout T k dil]   -> gpurun_out/pmc_conv_split_{a,b,c}.json
exec < /dev/null
export TMPDIR=/tmp
ARGS="${@:-32 128 128 5000 7 3}"
mkdir -p gpurun_out
timeout 100 python tools/conv_micro.py $ARGS 20 2>&1 | tail -1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  tag=$(echo a b c | cut -d' ' -f$((i+1))); i=$((i+1))
  rm -rf gpurun_out/pmc5
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc5 -o p -- python tools/conv_micro.py $ARGS 3 > gpurun_out/pmc_conv_split_$tag.log 2>&1
  echo "set $tag rc=$?"
  timeout 60 python tools/pmc_summary.py gpurun_out/pmc5 gpurun_out/pmc_conv_split_$tag.json | grep split | head -2
done
rm -rf gpurun_out/pmc5
