#!/bin/bash
# round 6: knock-outs of the ResBlock-unit kernel at the narrow stages: 1 = weight ring never refilled, 2 = B fragments read once per chunk, 3 = both
export SPAIR_NOLEG=1 SPAIR_DS=1
for lib in mockingbird_amd/libmbhip.so build_variants/libmbhip_spd1.so build_variants/libmbhip_spd2.so build_variants/libmbhip_spd3.so; do
  echo "== MBHIP_LIB=$lib"
  MBHIP_LIB=$lib python tools/spair_bench.py 10 128,5000 64,20000 32,40000 2>&1 | grep pair_us | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['C'], r['k'], r['pair_us'])"
done
