#!/bin/bash
# GPU session B of round 2: every -m gpu test, smoke, the default bench line, rocprofv3 kernel stats of the bench
# command and of the Tacotron / ppg2mel loops, and the PMC traffic passes of the benchmarked configurations.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -6 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
tail -c 9000 gpurun_out/bench.log
rm -rf gpurun_out/prof_bench
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o bench -- python bench.py --no-cpu-baseline --no-wavernn-unbatched > gpurun_out/prof_bench.log 2>&1; echo "prof_bench rc=$?"
find gpurun_out/prof_bench -type f ! -name '*stats*' -delete
f=$(find gpurun_out/prof_bench -name '*kernel_stats*' | head -1); [ -n "$f" ] && cut -c1-170 "$f" | head -24
if [ "${1:-}" != "nopmc" ]; then bash tools/pmc_r02.sh; fi
