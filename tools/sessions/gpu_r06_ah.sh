#!/bin/bash
# round 6: launches cut into whole rounds + remainder: tests, A/B
python -m pytest tests/test_resblock_pair_split_gpu.py -x -q -m gpu 2>&1 | tail -3
for r in 1 2; do
for d in "" spair_split=0; do
  echo "== MBHIP_DIAG=$d"
  MBHIP_DIAG=$d python tools/gan_run.py hifigan f32 32 200 20 2>&1 | tail -1
  [ $r = 1 ] && MBHIP_DIAG=$d python tools/gan_run.py fregan f32 8 1000 10 2>&1 | tail -1
done; done
python -m pytest tests/test_gan_gpu.py tests/test_gan_rb2_gpu.py tests/test_vits_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -3
