#!/bin/bash
# round 5: where Tacotron generate's time outside the decoder loop goes (kernel trace of 6 generate calls)
exec < /dev/null
mkdir -p gpurun_out; export TMPDIR=/tmp
python tools/taco_gen_time.py 2>&1 | grep -v amdgpu.ids
rm -rf gpurun_out/prof_tgen
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tgen -o tg -- python tools/taco_gen_time.py > gpurun_out/r05_prof_tgen.log 2>&1
f=$(find gpurun_out/prof_tgen -name '*kernel_stats*' | head -1); cp "$f" gpurun_out/r05_taco_generate_kernel_stats.csv
cut -c1-150 gpurun_out/r05_taco_generate_kernel_stats.csv | head -40
rm -rf gpurun_out/prof_tgen
