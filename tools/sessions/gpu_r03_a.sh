#!/bin/bash
# GPU session A of round 3: the new production-vs-oracle WaveRNN tests, the resident pipelined kernel (tests + A/B), then
# the whole -m gpu suite.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_wavernn_gpu.py tests/test_env_switches_gpu.py -m gpu -q -x --timeout=300 -k "production or debug_noise or pipe or persistent" > gpurun_out/pytest_new.log 2>&1; echo "pytest_new rc=$?"
tail -25 gpurun_out/pytest_new.log
timeout 600 python tools/wrn_pipe_ab.py > gpurun_out/pipe_ab.log 2>&1; echo "pipe_ab rc=$?"
tail -c 6000 gpurun_out/pipe_ab.log
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -12 gpurun_out/pytest_gpu.log
