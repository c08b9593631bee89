#!/bin/bash
# round 6: gaps of one Tacotron generate (kernel-trace timeline, hipGraph replays as they run in the product)
export TMPDIR=/tmp
rm -rf gpurun_out/prof_q
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_q -o taco -- python tools/taco_gen_time.py > gpurun_out/r06_taco_gaps.log 2>&1
tail -2 gpurun_out/r06_taco_gaps.log
f=$(find gpurun_out/prof_q -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last generate: from the last embed_gather to the end
idx=[i for i,r in enumerate(rows) if 'embed_gather' in r['Kernel_Name']][-1]
seg=rows[idx:]
t0=int(seg[0]['Start_Timestamp']); t1=max(int(r['End_Timestamp']) for r in seg)
busy=0; cur_end=t0; gaps=[]
for r in seg:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    if s>cur_end:
        gaps.append(((s-cur_end)/1e3, r['Kernel_Name'][:60], (s-t0)/1e3))
    busy+=max(0,e-max(s,cur_end)); cur_end=max(cur_end,e)
print('generate span ms', (t1-t0)/1e6, 'busy ms', busy/1e6, 'kernels', len(seg))
gaps.sort(reverse=True)
print('largest gaps (us, before kernel, at us):')
for g in gaps[:14]: print('  %8.1f  %-60s @%9.1f' % g)
print('sum of gaps > 3 us:', sum(g[0] for g in gaps if g[0]>3), 'count', sum(1 for g in gaps if g[0]>3))
PY
rm -rf gpurun_out/prof_q
