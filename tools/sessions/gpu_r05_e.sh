#!/bin/bash
# round 5 session: fused Tacotron front (taco_front_kernel): parity tests, then the same-box A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_tacotron_gpu.py -x -q -m gpu > gpurun_out/r05_taco_front_tests.log 2>&1
tail -5 gpurun_out/r05_taco_front_tests.log
timeout 600 python tools/taco_front_ab.py gpurun_out/r05_taco_front_ab.json 2>&1 | tail -8
