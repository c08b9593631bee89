#!/bin/bash
# GAN fp16 path: pair/gan parity tests, timing, per-kernel rocprofv3 stats (hifigan B=32 F=200).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "${1:-}" != "notest" ]; then
timeout 600 python -m pytest tests/test_resblock_pair_gpu.py tests/test_conv1d_f16_gpu.py tests/test_gan_gpu.py -m gpu -q 2>&1 | tail -4
fi
timeout 120 python tools/gan_run.py hifigan f16 32 200 10 2>&1 | grep batch
timeout 120 python tools/gan_run.py fregan f16 8 3000 3 2>&1 | grep batch
rm -rf gpurun_out/prof_f16
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_f16 -o gan -- python tools/gan_run.py hifigan f16 32 200 5 > gpurun_out/prof_f16.log 2>&1; echo "prof rc=$?"
find gpurun_out/prof_f16 -type f ! -name '*stats*' ! -name '*kernel_trace*' -delete
f=$(find gpurun_out/prof_f16 -name '*kernel_stats*' | head -1); head -12 "$f" | cut -c1-160
