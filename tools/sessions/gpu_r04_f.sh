#!/bin/bash
mkdir -p gpurun_out/r04f
timeout 900 python -m pytest tests/test_wavernn_gpu.py -x -q -m gpu -k "batch or production_pipe16" > gpurun_out/r04f/pytest_batch.log 2>&1; tail -12 gpurun_out/r04f/pytest_batch.log
timeout 600 python tools/wrn_batch32_ab.py > gpurun_out/r04f/batch32_ab.log 2>&1; tail -5 gpurun_out/r04f/batch32_ab.log
