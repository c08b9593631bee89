#!/bin/bash
# HBM counters of the ppg2mel decoder step (bench object ppg2mel, batch 1, T_enc = 200): separate FETCH_SIZE / WRITE_SIZE passes,
# eager launches (MBHIP_NO_GRAPH=1: counter mode crashes on hipGraph replays) -> gpurun_out/r03_pmc_ppg2mel.json
export TMPDIR=/tmp MBHIP_NO_GRAPH=1
mkdir -p gpurun_out
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc4_tmp
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/pmc4_tmp -o p -- python tools/ppg_run.py 1 > gpurun_out/pmc4_ppg_${ctr}.log 2>&1
  echo "$ctr rc=$?"
  python tools/pmc_summary.py gpurun_out/pmc4_tmp gpurun_out/pmc4_ppg_${ctr}.json | grep "ppg_" | head -8
done
rm -rf gpurun_out/pmc4_tmp
python - <<'PY'
import json, os
f = json.load(open("gpurun_out/pmc4_ppg_FETCH_SIZE.json")); w = json.load(open("gpurun_out/pmc4_ppg_WRITE_SIZE.json"))
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two separate passes) -- python tools/ppg_run.py 1 (bench object ppg2mel: "
                 "batch 1, T_enc = 200, 3 x 400 decoder steps), MBHIP_NO_GRAPH=1; KB per dispatch, mean over all dispatches; FETCH_SIZE doubled per "
                 "MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B)", "kernels": {}}
tot = 0.0
for k in f:
    if "ppg_" in k and k in w:
        fe, wr, n = f[k]["FETCH_SIZE"]["mean_per_dispatch"], w[k]["WRITE_SIZE"]["mean_per_dispatch"], f[k]["FETCH_SIZE"]["dispatches"]
        b = (2.0 * fe + wr) * 1024.0
        out["kernels"][k[:110]] = {"FETCH_SIZE_KB": fe, "WRITE_SIZE_KB": wr, "dispatches": n, "hbm_bytes_per_launch": b}
        tot += b * n
steps = 3 * 400
out["step_hbm_bytes_per_launch"] = tot / steps   # every launch of a step, summed
json.dump(out, open("gpurun_out/r03_pmc_ppg2mel.json", "w"), indent=1)
print("ppg2mel HBM bytes per step", out["step_hbm_bytes_per_launch"])
PY
