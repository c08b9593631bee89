#!/bin/bash
# GPU session F of round 3: the whole GPU suite, smoke, the bench line, rocprofv3 kernel stats of the bench command
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/pytest_gpu_f.log 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/pytest_gpu_f.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_f.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke_f.log
timeout 900 python bench.py > gpurun_out/bench_f.log 2>&1; echo "bench rc=$?"
tail -c 600 gpurun_out/bench_f.log
if [ "${1:-}" = "prof" ]; then
rm -rf gpurun_out/prof_bench_f
# (without the MOL object: its launches carry the headline kernel's name, the per-kernel average would mix the two)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench_f -o b -- python bench.py --no-wavernn-mol > gpurun_out/prof_bench_f.log 2>&1; echo "prof rc=$?"
find gpurun_out/prof_bench_f -type f ! -name '*kernel_stats*' -delete
fi
