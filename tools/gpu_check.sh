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
# HBM traffic counters: separate passes, no tracing domains mixed in (MI355X_MICROARCH.md: FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2)
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-hifigan --frames 200 > gpurun_out/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-hifigan --frames 200 > gpurun_out/pmc_write.log 2>&1; echo "pmc write rc=$?"
python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_fetch_summary.json
python tools/pmc_summary.py gpurun_out/pmc_write gpurun_out/pmc_write_summary.json
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
tail -3 gpurun_out/pytest_gpu.log
