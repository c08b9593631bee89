#!/bin/bash
# round 6: kernel stats of the bench command + kernel traces of one Tacotron generate and one fp32-result HiFi-GAN forward (per-kernel times)
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/sessions/gpu_r06_prof.sh prof
for w in taco gan; do
  if [ $w = taco ]; then cmd="python tools/taco_run.py 3"; else cmd="python tools/gan_run.py hifigan f32 32 200 3"; fi
  rm -rf gpurun_out/prof_$w
  MBHIP_NO_GRAPH=$([ $w = taco ] && echo 1 || echo 0) timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$w -o $w -- $cmd > gpurun_out/r06_prof_$w.log 2>&1
  f=$(find gpurun_out/prof_$w -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r06_${w}_kernel_stats.csv; rm -rf gpurun_out/prof_$w
  python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/r06_${w}_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('$w total ms', tot/1e6)
for r in rows[:28]: print('%9.1f us x %5s = %8.2f ms  %s' % (float(r['AverageNs'])/1e3, r['Calls'], float(r['TotalDurationNs'])/1e6, r['Name'][:110]))
PY
done
