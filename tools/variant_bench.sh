#!/bin/bash
# Batch-32 WaveRNN step time for variant libraries built by tools/build_variant.sh.
# usage: tools/variant_bench.sh "<variant> [ENV=value ...]" ...

for spec in "$@"; do
set -- $spec; v=$1; shift
env "$@" MBHIP_LIB=$PWD/build_variants/libmbhip_$v.so timeout 300 python bench.py --no-hifigan --no-tacotron --no-ppg2mel --no-cpu-baseline --steps 1 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); b=r['wavernn_batch32']; print('$spec batch32: %.1f us/step' % b['us_per_time_step'])"
done
