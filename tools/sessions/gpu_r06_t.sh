#!/bin/bash
python -m pytest tests/test_conv_split_tm_gpu.py tests/test_gan_gpu.py -x -q 2>&1 | tail -3
python -m pytest tests/test_tacotron_gpu.py -x -q 2>&1 | tail -3
for sh in "512 1024 400 1 32 split" "2560 512 400 3 32 split"; do
  echo "== $sh"
  MBHIP_LIB=$PWD/build_variants/libmbhip_ctmtrace.so python tools/ctm_trace.py $sh 2>&1 | grep -v amdgpu.ids | tail -5
done
python tools/taco_gen_time.py 2>&1 | tail -2
MBHIP_DIAG=ctm_jc1 python tools/taco_gen_time.py 2>&1 | tail -2
python tools/gan_run.py hifigan f32 32 200 5 2>&1 | tail -1
MBHIP_DIAG=ctm_jc1 python tools/gan_run.py hifigan f32 32 200 5 2>&1 | tail -1
