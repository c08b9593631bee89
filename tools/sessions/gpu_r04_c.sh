#!/bin/bash
# round 4, session C: first light of wavernn_pipe16.h (oracle replay + speed + marks), conv split with the scaled residual
mkdir -p gpurun_out/r04c
python -m pytest tests/test_conv1d_gpu.py -x -q -m gpu -s > gpurun_out/r04c/pytest_conv.log 2>&1; tail -3 gpurun_out/r04c/pytest_conv.log
timeout 900 python -m pytest tests/test_wavernn_gpu.py -x -q -m gpu -k "production_pipe16 or production_fast_chain" -s > gpurun_out/r04c/pytest_wq16.log 2>&1; tail -15 gpurun_out/r04c/pytest_wq16.log
timeout 300 python tools/wrn_pipe_sweep.py 17 > gpurun_out/r04c/wq16_speed.log 2>&1; cat gpurun_out/r04c/wq16_speed.log
MBHIP_WAVERNN_RESIDENT=exact timeout 300 python tools/wrn_pipe_sweep.py 1,17 > gpurun_out/r04c/wq_exact_speed.log 2>&1; cat gpurun_out/r04c/wq_exact_speed.log
WQ_AB_CASES=configs1_23_folds timeout 300 python tools/wrn_pipe_ab.py gpurun_out/r04c/ab_wq16.json > gpurun_out/r04c/ab_wq16.log 2>&1
WQ_AB_CASES=configs1_23_folds WQ_AB_EXACT=1 timeout 300 python tools/wrn_pipe_ab.py gpurun_out/r04c/ab_exact.json > gpurun_out/r04c/ab_exact.log 2>&1
tail -1 gpurun_out/r04c/ab_wq16.log | cut -c1-300
