#!/bin/bash
# fp16 GAN path: parity tests, timing, per-kernel rocprofv3 stats.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv1d_f16_gpu.py tests/test_gan_gpu.py -m gpu -q -s > gpurun_out/pytest_f16.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|f16 parity|f16 vs|FAILED|Error|error" gpurun_out/pytest_f16.log | head -60
timeout 120 python tools/gan_run.py hifigan f32 32 200 3 2>&1 | tail -1
timeout 120 python tools/gan_run.py hifigan f16 32 200 10 2>&1 | tail -1
timeout 120 python tools/gan_run.py fregan f16 8 3000 3 2>&1 | tail -1
rm -rf gpurun_out/prof_f16
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_f16 -o gan -- python tools/gan_run.py hifigan f16 32 200 5 > gpurun_out/prof_f16.log 2>&1; echo "prof rc=$?"
find gpurun_out/prof_f16 -type f ! -name '*stats*' ! -name '*kernel_trace*' -delete
f=$(find gpurun_out/prof_f16 -name '*kernel_stats*' | head -1); head -12 "$f"
