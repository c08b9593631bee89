#!/bin/bash
# round 6 evidence on the final kernels: kernel stats of the bench command, matrix-pipe occupancy + vector instructions per MFMA and
# HBM traffic of the fp32-result HiFi-GAN forward (separate counter passes); the summaries land in gpurun_out/ (copy into profiles/)
mkdir -p gpurun_out
bash tools/sessions/gpu_r06_prof.sh prof
export MBHIP_DIAG=gan_one_stream   # (counter mode serialises the launches anyway)
bash tools/pmc_gan.sh f32 2>&1 | tail -24
MB_PMC_ROUND=r06 MB_PMC_DTYPE=f32 python tools/pmc_gan_summary.py 2>&1 | tail -12
MB_PMC_ROUND=r06 bash tools/pmc_r02.sh hifigan_f32 2>&1 | tail -6
cp profiles/r06_hifigan_f32_mfma_util.json profiles/r06_pmc_hifigan_f32.json gpurun_out/ 2>/dev/null
