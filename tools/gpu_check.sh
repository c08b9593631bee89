#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprofv3 kernel stats. Outputs under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/smoke.log
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
tail -c 6000 gpurun_out/bench.log
rm -rf gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_prof.log 2>&1; echo "prof rc=$?"
find gpurun_out/prof -name '*kernel_stats*' | head
# keep only stats csvs (traces are large)
find gpurun_out/prof -type f ! -name '*stats*' -delete
tail -5 gpurun_out/pytest_gpu.log
