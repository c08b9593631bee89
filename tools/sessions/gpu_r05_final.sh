#!/bin/bash
# Round-5 measurement session, in stages (usage: bash tools/sessions/gpu_r05_final.sh <tag> <stage>...):
#   suite  every -m gpu test, __graft_entry__.smoke(), the default bench line
#   prof   rocprofv3 --kernel-trace --stats of the bench command -> profiles/r05_bench_kernel_stats.csv
#   pmcw   HBM traffic + SQ counters of the resident WaveRNN kernel -> profiles/r05_pmc_wavernn.json, r05_wavernn_pipe16_sq_counters.json
#   pmc    HBM traffic of the Tacotron iteration, both HiFi-GAN precisions, Fre-GAN fp16, ppg2mel -> profiles/r05_pmc_*.json
exec < /dev/null
set -u
TAG=${1:-a}; shift
mkdir -p gpurun_out profiles
export TMPDIR=/tmp
for stage in "$@"; do
case $stage in
suite)
  timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -s > gpurun_out/r05_pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r05_pytest_gpu_$TAG.log
  grep -v "amdgpu.ids" gpurun_out/r05_pytest_gpu_$TAG.log | tail -4
  timeout 300 python __graft_entry__.py --smoke > gpurun_out/r05_smoke_$TAG.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/r05_smoke_$TAG.log
  timeout 900 python bench.py > gpurun_out/r05_bench_$TAG.log 2>&1; echo "bench rc=$?"
  grep "^{" gpurun_out/r05_bench_$TAG.log | head -c 1200; echo ;;
prof)
  rm -rf gpurun_out/prof_bench5
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench5 -o bench -- python bench.py --no-cpu-baseline --no-wavernn-unbatched --no-wavernn-mol > gpurun_out/r05_prof_bench_$TAG.log 2>&1; echo "prof_bench rc=$?"
  f=$(find gpurun_out/prof_bench5 -name '*kernel_stats*' | head -1); if [ -n "$f" ]; then cp "$f" profiles/r05_bench_kernel_stats.csv; cp "$f" gpurun_out/r05_bench_kernel_stats.csv; cut -c1-170 "$f" | head -14; fi
  rm -rf gpurun_out/prof_bench5 ;;
pmcw)
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc4_tmp
    timeout -k 5 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/pmc4_tmp -o p -- python tools/wrn_run.py 1000 1 > gpurun_out/pmc4_wavernn_${ctr}.log 2>&1
    echo "$ctr rc=$?"
    timeout 100 python tools/pmc_summary.py gpurun_out/pmc4_tmp gpurun_out/pmc4_wavernn_${ctr}.json | grep "wf_" | head -3
  done
  rm -rf gpurun_out/pmc4_tmp
  bash tools/pmc_wavernn_sq.sh
  timeout 60 python tools/pmc_wavernn_r05_json.py
  cp profiles/r05_pmc_wavernn.json profiles/r05_wavernn_pipe16_sq_counters.json gpurun_out/ 2>/dev/null ;;
pmc)
  MB_PMC_ROUND=r05 timeout 1500 bash tools/pmc_r02.sh tacotron hifigan hifigan_f32 fregan
  timeout 600 bash tools/pmc_r03_ppg.sh && cp gpurun_out/r03_pmc_ppg2mel.json profiles/r05_pmc_ppg2mel.json
  cp profiles/r05_pmc_*.json gpurun_out/ 2>/dev/null ;;
esac
done
