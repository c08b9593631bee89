#!/bin/bash
# GPU session B of round 3: every -m gpu test, smoke, the default bench line, rocprofv3 kernel stats of the bench command,
# PMC traffic passes of the resident WaveRNN kernel.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
tail -c 3000 gpurun_out/bench.log
if [ "${1:-}" != "noprof" ]; then
rm -rf gpurun_out/prof_bench
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o bench -- python bench.py --no-cpu-baseline --no-wavernn-unbatched > gpurun_out/prof_bench.log 2>&1; echo "prof_bench rc=$?"
find gpurun_out/prof_bench -type f ! -name '*stats*' -delete
f=$(find gpurun_out/prof_bench -name '*kernel_stats*' | head -1); [ -n "$f" ] && cut -c1-170 "$f" | head -16
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc3_tmp
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/pmc3_tmp -o p -- python tools/wrn_run.py 1000 1 > gpurun_out/pmc3_wavernn_${ctr}.log 2>&1
  echo "$ctr rc=$?"; tail -2 gpurun_out/pmc3_wavernn_${ctr}.log
  python tools/pmc_summary.py gpurun_out/pmc3_tmp gpurun_out/pmc3_wavernn_${ctr}.json | grep "wf_" | head -3
done
rm -rf gpurun_out/pmc3_tmp
python tools/pmc_wavernn_r03_json.py
fi
