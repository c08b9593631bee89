#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_conv_split_tm_gpu.py tests/test_resblock_pair_split_gpu.py -x -q 2>&1 | tail -12
python -m pytest tests/test_gan_gpu.py tests/test_vits_gpu.py tests/test_pipeline_gpu.py -x -q 2>&1 | tail -12
python tools/gan_run.py hifigan f32 32 200 5 2>&1 | tail -1
python tools/gan_run.py fregan f32 8 1000 3 2>&1 | tail -1
MBHIP_DIAG=gan_tm_pairs_only python tools/gan_run.py hifigan f32 32 200 5 2>&1 | tail -1
