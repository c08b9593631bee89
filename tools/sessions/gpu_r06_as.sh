#!/bin/bash
# round 6: 64 channels as one 64-channel chunk (weight ring of eight k-steps): tests + A/B
python -m pytest tests/test_resblock_pair_split_gpu.py -x -q -m gpu 2>&1 | tail -2
export SPAIR_NOLEG=1 SPAIR_DS=1,5
for lib in mockingbird_amd/libmbhip.so build_variants/libmbhip_ck32.so mockingbird_amd/libmbhip.so build_variants/libmbhip_ck32.so; do
  echo "== MBHIP_LIB=$lib"
  MBHIP_LIB=$lib python tools/spair_bench.py 10 64,20000 2>&1 | grep pair_us | python -c "
import sys, json
print([(r['k'], r['d'], r['pair_us']) for r in map(json.loads, sys.stdin) if 'k' in r])"
  MBHIP_LIB=$lib python tools/gan_run.py hifigan f32 32 200 20 2>&1 | tail -1
done
