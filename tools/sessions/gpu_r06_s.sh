#!/bin/bash
for sh in "512 1024 400 1 32 split" "512 1024 400 1 32" "2560 512 400 3 32 split" "80 2560 400 5 32" "512 768 400 1 32 split" "256 640 5000 3 32"; do
  echo "== $sh"
  MBHIP_LIB=$PWD/build_variants/libmbhip_ctmtrace.so python tools/ctm_trace.py $sh 2>&1 | grep -v amdgpu.ids | tail -5
done
python tools/gan_run.py hifigan f32 32 200 5 2>&1 | tail -1
python tools/taco_gen_time.py 2>&1 | tail -2
