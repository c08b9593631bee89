#!/bin/bash
# round 4, session D: pipe16 with fast gates + one-barrier key decode: oracle replay, speed, marks; full wavernn + env-switch suites
mkdir -p gpurun_out/r04d
timeout 900 python -m pytest tests/test_wavernn_gpu.py -x -q -m gpu -k "production_pipe16 or production_fast_chain" > gpurun_out/r04d/pytest_wq16.log 2>&1; tail -4 gpurun_out/r04d/pytest_wq16.log
timeout 300 python tools/wrn_pipe_sweep.py 17 > gpurun_out/r04d/wq16_speed.log 2>&1; cat gpurun_out/r04d/wq16_speed.log
WQ_AB_CASES=configs1_23_folds timeout 300 python tools/wrn_pipe_ab.py gpurun_out/r04d/ab_wq16.json > gpurun_out/r04d/ab_wq16.log 2>&1
timeout 1500 python -m pytest tests/test_wavernn_gpu.py tests/test_env_switches_gpu.py -q -m gpu > gpurun_out/r04d/pytest_wavernn_all.log 2>&1; tail -8 gpurun_out/r04d/pytest_wavernn_all.log
