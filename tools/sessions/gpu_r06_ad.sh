#!/bin/bash
# round 6: per-kernel times of the fp16 storage path (HiFi-GAN 32 x 200, Fre-GAN 8 x 3000)
export TMPDIR=/tmp
for w in "hifigan f16 32 200" "fregan f16 8 3000"; do
  rm -rf gpurun_out/prof_f
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_f -o g -- python tools/gan_run.py $w 3 > gpurun_out/r06_prof_f.log 2>&1
  tail -1 gpurun_out/r06_prof_f.log
  f=$(find gpurun_out/prof_f -name "*kernel_stats.csv" | head -1)
  python - "$f" <<PY
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('$w total ms per forward', tot/1e6/4)
for r in rows[:16]: print('%9.1f us x %5s = %8.2f ms  %s' % (float(r['AverageNs'])/1e3, r['Calls'], float(r['TotalDurationNs'])/1e6/4, r['Name'][:100]))
PY
  rm -rf gpurun_out/prof_f
done
