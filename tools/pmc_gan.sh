#!/bin/bash
# MFMA utilisation of the HiFi-GAN fp16 pass (north_star: "MFMA utilisation on the HiFi-GAN convs against gfx950
# peak"): rocprofv3 counter mode, one counter group per pass, per-kernel means -> gpurun_out/pmc_gan_*.json
# usage: bash tools/pmc_gan.sh [f16|f32]   (then MB_PMC_ROUND=r04 MB_PMC_DTYPE=f32 python tools/pmc_gan_summary.py)
exec < /dev/null
DT=${1:-f16}
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/pmc_gan_*.json
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  tag=$(echo $grp | tr ' ' '+')
  rm -rf gpurun_out/pmc_gan_tmp
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmc_gan_tmp -o p -- python tools/gan_run.py hifigan $DT 32 200 3 > gpurun_out/pmc_gan.log 2>&1
  echo "$tag rc=$?"
  python tools/pmc_summary.py gpurun_out/pmc_gan_tmp gpurun_out/pmc_gan_$tag.json | grep -E "resblock_pair|resblock_stage|conv1d_f16|conv1d_split|conv_split_tm" | head -12
done
rm -rf gpurun_out/pmc_gan_tmp
