#!/bin/bash
# round 6 session B: where the split pair kernel's time goes (diagnostics variants, results wrong by construction)
mkdir -p gpurun_out
for v in 0 1 2 3 4 8 16 27 32; do
  echo "== variant SPAIR_DBG=$v" >> gpurun_out/r06_b_variants.log
  MBHIP_LIB=$PWD/build_variants/libmbhip_sp$v.so SPAIR_NOLEG=1 SPAIR_KS=3,11 SPAIR_DS=1 timeout 300 python tools/spair_bench.py 10 2>&1 | grep pair_us >> gpurun_out/r06_b_variants.log
done
cat gpurun_out/r06_b_variants.log
