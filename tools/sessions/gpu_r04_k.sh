#!/bin/bash
# per-launch durations of the fp32-result HiFi-GAN forward (32 x 200 frames), grouped by kernel + grid
exec < /dev/null
mkdir -p gpurun_out/r04k
timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04k/prof -o p -- python tools/gan_run.py hifigan f32 32 200 3 > gpurun_out/r04k/prof.log 2>&1
grep -v "^W2\|^E2\|^I2" gpurun_out/r04k/prof.log | tail -2
f=$(find gpurun_out/r04k/prof -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
g = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"][:70]
    key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    g.setdefault(key, []).append(d)
tot = sum(sum(v) for v in g.values())
out = []
for k, v in g.items():
    out.append((sum(v), k, len(v), sum(v) / len(v)))
out.sort(reverse=True)
print("total us", tot, "per forward (4 forwards incl. warm-up)", tot / 4)
for s, k, n, a in out[:40]:
    print(f"{s/4:9.1f} us/fwd  n/fwd={n/4:5.1f} avg={a:8.1f}  {k}")
PY
fi
rm -rf gpurun_out/r04k/prof
