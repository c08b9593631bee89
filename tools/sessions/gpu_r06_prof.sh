#!/bin/bash
# round 6 evidence: rocprofv3 kernel stats of the bench command, matrix-pipe occupancy + vector instructions per MFMA of the fp32-result
# HiFi-GAN forward (five separate counter passes), HBM traffic of the same forward (FETCH_SIZE / WRITE_SIZE passes)
mkdir -p gpurun_out
export TMPDIR=/tmp
for w in "$@"; do case $w in
 prof)
  timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench6 -o bench -- python bench.py --no-cpu-baseline --no-wavernn-unbatched --no-wavernn-mol > gpurun_out/r06_prof_bench.log 2>&1; echo "prof_bench rc=$?"
  f=$(find gpurun_out/prof_bench6 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r06_bench_kernel_stats.csv; rm -rf gpurun_out/prof_bench6
  head -12 gpurun_out/r06_bench_kernel_stats.csv | cut -c1-160 ;;
 util)
  bash tools/pmc_gan.sh f32 2>&1 | tail -30 ;;
 traffic)
  MB_PMC_ROUND=r06 bash tools/pmc_r02.sh hifigan_f32 2>&1 | tail -6 ;;
esac; done
