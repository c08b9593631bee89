#!/bin/bash
# round 6: second accumulator set instead of the third weight image (one-M-tile instances): tests + A/B
python -m pytest tests/test_resblock_pair_split_gpu.py tests/test_conv_split_tm_gpu.py -x -q -m gpu 2>&1 | tail -3
export SPAIR_NOLEG=1 SPAIR_DS=1
for lib in mockingbird_amd/libmbhip.so build_variants/libmbhip_nodual.so mockingbird_amd/libmbhip.so build_variants/libmbhip_nodual.so; do
  echo "== MBHIP_LIB=$lib"
  MBHIP_LIB=$lib python tools/spair_bench.py 10 128,5000 64,20000 32,40000 2>&1 | tail -1
  MBHIP_LIB=$lib python tools/gan_run.py hifigan f32 32 200 20 2>&1 | tail -1
done
python -m pytest tests/test_gan_gpu.py tests/test_gan_rb2_gpu.py tests/test_vits_gpu.py tests/test_pipeline_gpu.py tests/test_tacotron_gpu.py -x -q -m gpu 2>&1 | tail -3
