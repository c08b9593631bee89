#!/bin/bash
# Round-4 measurement session: every -m gpu test, smoke, the default bench line, rocprofv3 kernel stats of the bench command,
# PMC traffic passes of the resident WaveRNN kernel.  usage: bash tools/sessions/gpu_r04_final.sh [tag] [noprof]
exec < /dev/null
set -u
TAG=${1:-a}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -s > gpurun_out/r04_pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r04_pytest_gpu_$TAG.log
grep -v "amdgpu.ids" gpurun_out/r04_pytest_gpu_$TAG.log | tail -4
timeout 300 python __graft_entry__.py --smoke > gpurun_out/r04_smoke_$TAG.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/r04_smoke_$TAG.log
timeout 900 python bench.py > gpurun_out/r04_bench_$TAG.log 2>&1; echo "bench rc=$?"
grep "^{" gpurun_out/r04_bench_$TAG.log | head -c 1500; echo
if [ "${2:-}" != "noprof" ]; then
rm -rf gpurun_out/prof_bench4
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench4 -o bench -- python bench.py --no-cpu-baseline --no-wavernn-unbatched --no-wavernn-mol > gpurun_out/r04_prof_bench_$TAG.log 2>&1; echo "prof_bench rc=$?"
f=$(find gpurun_out/prof_bench4 -name '*kernel_stats*' | head -1); if [ -n "$f" ]; then cp "$f" gpurun_out/r04_bench_kernel_stats.csv; cut -c1-170 "$f" | head -14; fi
rm -rf gpurun_out/prof_bench4
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc4_tmp
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/pmc4_tmp -o p -- python tools/wrn_run.py 1000 1 > gpurun_out/pmc4_wavernn_${ctr}.log 2>&1
  echo "$ctr rc=$?"
  timeout 100 python tools/pmc_summary.py gpurun_out/pmc4_tmp gpurun_out/pmc4_wavernn_${ctr}.json | grep "wf_" | head -3
done
rm -rf gpurun_out/pmc4_tmp
timeout 60 python tools/pmc_wavernn_r04_json.py
# the Tacotron iteration's kernels (hardware gate functions since round 4): profiles/r04_pmc_tacotron.json
MB_PMC_ROUND=r04 timeout 1200 bash tools/pmc_r02.sh tacotron hifigan_f32
fi
