#!/bin/bash
# round 6: do the ResBlock chains on branch streams overlap?  kernel-trace timeline of one forward
export TMPDIR=/tmp
rm -rf gpurun_out/prof_y
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_y -o gan -- python tools/gan_run.py hifigan f32 32 200 2 > /dev/null 2>&1
f=$(find gpurun_out/prof_y -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'f32_transpose' in r['Kernel_Name']][-1]
t0=int(rows[idx]['Start_Timestamp'])
print(list(rows[0].keys()))
for r in rows[idx:idx+48]:
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} .. {(int(r['End_Timestamp'])-t0)/1e3:9.1f} us  q{r.get('Queue_Id','?'):>3s}  {r['Kernel_Name'][:80]}")
PY
rm -rf gpurun_out/prof_y
