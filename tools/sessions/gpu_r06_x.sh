#!/bin/bash
# round 6: ResBlock chains on branch streams, A/B inside one call
for r in 1 2; do
for d in "" gan_sum_fwd gan_fork_rounds=4 gan_fork_rounds=100 gan_one_stream; do
  echo "== MBHIP_DIAG=$d"
  MBHIP_DIAG=$d python tools/gan_run.py hifigan f32 32 200 20 2>&1 | tail -1
  [ $r = 1 ] && MBHIP_DIAG=$d python tools/gan_run.py fregan f32 8 1000 10 2>&1 | tail -1
done; done
bash tools/sessions/gpu_r06_y.sh 2>&1 | grep -v "^\[" | head -24
python -m pytest tests/test_gan_gpu.py tests/test_gan_rb2_gpu.py tests/test_vits_gpu.py tests/test_pipeline_gpu.py tests/test_resblock_stage_f32_gpu.py -x -q -m gpu 2>&1 | tail -4
