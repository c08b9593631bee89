#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_j -o gan -- python tools/gan_run.py hifigan f32 32 200 5 > gpurun_out/r06_j_prof.log 2>&1
f=$(find gpurun_out/prof_j -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r06_j_hifigan_f32_kernel_stats.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r06_j_hifigan_f32_kernel_stats.csv')))
tot=0
for r in rows:
    n=int(r['Calls']); t=float(r['TotalDurationNs'])
    print(f"{r['Name'][:110]:110s} calls/fwd {n/6:6.1f}  us/fwd {t/6/1e3:8.1f}  avg {float(r['AverageNs'])/1e3:8.1f}")
    tot+=t
print('total us/fwd', tot/6/1e3)
PY
rm -rf gpurun_out/prof_j
