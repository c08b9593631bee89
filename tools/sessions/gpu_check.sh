#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprofv3 kernel stats + HBM-traffic counters.
# Everything lands under gpurun_out/; copy what should be judged into profiles/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/smoke.log
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
tail -c 6000 gpurun_out/bench.log
rm -rf gpurun_out/prof gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_prof.log 2>&1; echo "prof rc=$?"
find gpurun_out/prof -name '*kernel_stats*' | head
find gpurun_out/prof -type f ! -name '*stats*' -delete
# HBM traffic counters: tools/pmc_run.sh (counter mode segfaults on hipGraph replays -> MBHIP_NO_GRAPH=1 there)
if [ -z "${SKIP_PMC:-}" ]; then
bash tools/pmc_run.sh
python tools/pmc_wavernn_json.py
fi
tail -3 gpurun_out/pytest_gpu.log
