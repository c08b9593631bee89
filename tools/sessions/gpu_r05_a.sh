#!/bin/bash
# Round-5 session A: the WaveRNN suite first (the new full-length replays), the headline A/B against round 4's library on the same box,
# then the rest of the -m gpu suite and a short bench.
exec < /dev/null
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_wavernn_gpu.py tests/test_env_switches_gpu.py tests/test_host_logic.py -m gpu -q --timeout=600 -x -s > gpurun_out/r05_pytest_wavernn_a.log 2>&1; echo "pytest wavernn rc=$?"
grep -v "amdgpu.ids" gpurun_out/r05_pytest_wavernn_a.log | grep -i "passed\|failed\|error\|replay\]" | tail -12
MBHIP_LIB=$PWD/build_variants/libmbhip_r04.so timeout 300 python tools/wrn_ab_r05.py r04 > gpurun_out/r05_ab_r04.log 2>&1; echo "ab r04 rc=$?"
timeout 300 python tools/wrn_ab_r05.py r05 > gpurun_out/r05_ab_r05.log 2>&1; echo "ab r05 rc=$?"
MBHIP_LIB=$PWD/build_variants/libmbhip_r04.so timeout 300 python tools/wrn_ab_r05.py r04b quick > gpurun_out/r05_ab_r04b.log 2>&1
timeout 300 python tools/wrn_ab_r05.py r05b quick > gpurun_out/r05_ab_r05b.log 2>&1
grep -h "^configs1\|folds" gpurun_out/r05_ab_r04.log gpurun_out/r05_ab_r05.log gpurun_out/r05_ab_r04b.log gpurun_out/r05_ab_r05b.log | cut -c1-260
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 --deselect tests/test_wavernn_gpu.py > gpurun_out/r05_pytest_rest_a.log 2>&1; echo "pytest rest rc=$?"
grep -v "amdgpu.ids" gpurun_out/r05_pytest_rest_a.log | tail -4
timeout 600 python bench.py --no-e2e > gpurun_out/r05_bench_a.log 2>&1; echo "bench rc=$?"
grep "^{" gpurun_out/r05_bench_a.log | head -c 3000; echo
