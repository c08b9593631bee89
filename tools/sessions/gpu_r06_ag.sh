#!/bin/bash
# round 6: pipelined loader of the fp16 ResBlock-unit kernel: tests + A/B against the previous library
python -m pytest tests/test_resblock_pair_gpu.py tests/test_gan_gpu.py -x -q -m gpu 2>&1 | tail -3
for r in 1 2; do
for lib in mockingbird_amd/libmbhip.so build_variants/libmbhip_nopipe.so; do
  echo "== $lib"
  MBHIP_LIB=$lib python tools/gan_run.py hifigan f16 32 200 20 2>&1 | tail -1
  MBHIP_LIB=$lib python tools/gan_run.py fregan f16 8 3000 10 2>&1 | tail -1
done; done
