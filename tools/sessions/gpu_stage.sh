#!/bin/bash
# one-launch ResBlock group of the narrow GAN stages: parity tests, GAN tests, timing, kernel stats
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_resblock_stage_gpu.py -m gpu -q -x --timeout=120 2>&1 | tail -15
if [ "${1:-}" != "quick" ]; then
timeout 600 python -m pytest tests/test_resblock_pair_gpu.py tests/test_gan_gpu.py -m gpu -q --timeout=300 2>&1 | tail -5
fi
timeout 120 python tools/gan_run.py hifigan f16 32 200 10 2>&1 | grep batch
timeout 120 python tools/gan_run.py fregan f16 8 3000 3 2>&1 | grep batch
rm -rf gpurun_out/prof_f16
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_f16 -o gan -- python tools/gan_run.py hifigan f16 32 200 5 > gpurun_out/prof_f16.log 2>&1; echo "prof rc=$?"
find gpurun_out/prof_f16 -type f ! -name '*stats*' -delete
f=$(find gpurun_out/prof_f16 -name '*kernel_stats*' | head -1); head -14 "$f" | cut -c1-150
rm -rf gpurun_out/prof_fre
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fre -o gan -- python tools/gan_run.py fregan f16 8 3000 3 > gpurun_out/prof_fre.log 2>&1; echo "prof rc=$?"
find gpurun_out/prof_fre -type f ! -name '*stats*' -delete
f=$(find gpurun_out/prof_fre -name '*kernel_stats*' | head -1); head -14 "$f" | cut -c1-150
