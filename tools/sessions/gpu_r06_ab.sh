#!/bin/bash
# round 6: per-kernel times of the fp32-result HiFi-GAN forward + the tests that run the time-major kernels
export TMPDIR=/tmp
for w in gan; do
  cmd="python tools/gan_run.py hifigan f32 32 200 3"
  rm -rf gpurun_out/prof_$w
  MBHIP_DIAG=gan_one_stream timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$w -o $w -- $cmd > gpurun_out/r06_prof_$w.log 2>&1
  f=$(find gpurun_out/prof_$w -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r06_${w}_kernel_stats.csv; rm -rf gpurun_out/prof_$w
  python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/r06_${w}_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('$w total ms', tot/1e6)
for r in rows[:14]: print('%9.1f us x %5s = %8.2f ms  %s' % (float(r['AverageNs'])/1e3, r['Calls'], float(r['TotalDurationNs'])/1e6, r['Name'][:110]))
PY
done
python -m pytest tests/test_gan_gpu.py tests/test_gan_rb2_gpu.py tests/test_vits_gpu.py tests/test_pipeline_gpu.py tests/test_resblock_stage_f32_gpu.py tests/test_tacotron_gpu.py tests/test_conv_split_tm_gpu.py tests/test_resblock_pair_split_gpu.py tests/test_env_switches_gpu.py -x -q -m gpu 2>&1 | tail -4
