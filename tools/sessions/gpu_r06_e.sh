#!/bin/bash
mkdir -p gpurun_out
for sh in "256 1000 3 1" "256 1000 11 1" "128 5000 3 1" "128 5000 11 1" "64 20000 3 1" "64 20000 11 1" "32 40000 3 1" "32 40000 11 1"; do
  echo "== $sh" >> gpurun_out/r06_e_trace.log
  MBHIP_LIB=$PWD/build_variants/libmbhip_sptrace.so python tools/spair_trace.py $sh 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_e_trace.log
done
cat gpurun_out/r06_e_trace.log
