#!/bin/bash
# GPU session A of round 2: parity (all -m gpu tests), smoke, Tacotron fast-vs-general timing + kernel stats, bench.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/smoke.log
timeout 600 python tools/taco_time.py > gpurun_out/taco_time.log 2>&1; echo "taco_time rc=$?"; cat gpurun_out/taco_time.log | tail -5
rm -rf gpurun_out/prof_taco
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_taco -o taco -- python tools/taco_run.py 2 > gpurun_out/prof_taco.log 2>&1; echo "prof_taco rc=$?"
find gpurun_out/prof_taco -type f ! -name '*stats*' -delete
f=$(find gpurun_out/prof_taco -name '*kernel_stats*' | head -1); [ -n "$f" ] && cut -c1-150 "$f" | head -16
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
tail -c 7000 gpurun_out/bench.log
