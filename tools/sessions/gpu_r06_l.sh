#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gan_rb2_gpu.py tests/test_pipeline_gpu.py tests/test_gan_gpu.py tests/test_vits_gpu.py -x -q -s 2>&1 | grep -v "^$" | tail -15 | tee gpurun_out/r06_l_pytest_a.log
python -m pytest "tests/test_tacotron_gpu.py::test_baseline_config2_full_length_vs_oracle" -x -q -s 2>&1 | grep -v "^$" | tail -8 | tee gpurun_out/r06_l_pytest_b.log
python -m pytest "tests/test_wavernn_gpu.py::test_production_batch32_full_size_vs_oracle" -x -q -s 2>&1 | grep -v "^$" | tail -6 | tee gpurun_out/r06_l_pytest_c.log
