#!/bin/bash
mkdir -p gpurun_out/r04e
timeout 900 python -m pytest tests/test_wavernn_gpu.py -x -q -m gpu -k "production_pipe16 or production_fast_chain" > gpurun_out/r04e/pytest_wq16.log 2>&1; tail -4 gpurun_out/r04e/pytest_wq16.log
timeout 600 python tools/wrn_groups.py > gpurun_out/r04e/wrn_groups.log 2>&1; cat gpurun_out/r04e/wrn_groups.log; cp gpurun_out/wrn_groups.json gpurun_out/r04e/
