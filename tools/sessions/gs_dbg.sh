#!/bin/bash
# resident GRU scan (gru_scan.h): kernel durations under each diagnostics bit (rocprofv3 kernel trace of one Tacotron generate)
export TMPDIR=/tmp
mkdir -p gpurun_out
for dbg in ${@:-0 1 2 4 8 15}; do
  rm -rf gpurun_out/prof_gs
  MBHIP_DIAG=gs_dbg=$dbg timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gs -o t -- python tools/taco_gen_time.py > gpurun_out/prof_gs.log 2>&1
  f=$(find gpurun_out/prof_gs -name '*kernel_stats*' | head -1)
  python - "$f" $dbg <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "gru_scan" in r["Name"] or "rnn_dual_gru" in r["Name"]:
        print(f"dbg={sys.argv[2]} {r['Name'][:48]:48s} calls={r['Calls']} avg_us={float(r['AverageNs'])/1e3:.1f}")
PY
done
