#!/bin/bash
# per-dispatch durations and grids of the fp32 conv kernel inside one Tacotron generate (B=32, 400 frames)
export TMPDIR=/tmp
rm -rf gpurun_out/prof_cc
timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_cc -o cc -- python tools/taco_run.py 1 > gpurun_out/prof_cc.log 2>&1
f=$(find gpurun_out/prof_cc -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = []
for r in rows:
    if "conv1d_mfma" in r["Kernel_Name"]:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        out.append((r["Kernel_Name"][10:45], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"], round(d, 1)))
for o in out: print(*o)
print("total us", sum(o[-1] for o in out), "calls", len(out))
PY
rm -rf gpurun_out/prof_cc
