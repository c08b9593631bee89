"""profiles/r03_pmc_wavernn.json from the two rocprofv3 --pmc passes of tools/sessions/gpu_r03_b.sh over tools/wrn_run.py
(BASELINE configs[1]: 23 folds x 9600 steps, the resident wf_pipe_kernel): HBM bytes per launch = 2 x FETCH_SIZE +
WRITE_SIZE (KB per dispatch; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note)."""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f = json.load(open(os.path.join(ROOT, "gpurun_out", "pmc3_wavernn_FETCH_SIZE.json")))
w = json.load(open(os.path.join(ROOT, "gpurun_out", "pmc3_wavernn_WRITE_SIZE.json")))
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two separate passes) -- python tools/wrn_run.py 1000 1 "
                 "(BASELINE configs[1]: mel 80x1000 -> 23 folds x 9600 steps, ONE resident launch of wf_pipe_kernel); KB per dispatch; "
                 "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B)", "kernels": {}}
for k in f:
    if k in w and "FETCH_SIZE" in f[k] and "WRITE_SIZE" in w[k]:
        fe, wr = f[k]["FETCH_SIZE"]["mean_per_dispatch"], w[k]["WRITE_SIZE"]["mean_per_dispatch"]
        out["kernels"][k[:100]] = {"FETCH_SIZE_KB": fe, "WRITE_SIZE_KB": wr, "dispatches": f[k]["FETCH_SIZE"]["dispatches"],
                                   "hbm_bytes_per_launch": (2.0 * fe + wr) * 1024.0}
        if "wf_pipe_kernel" in k:
            out["pipe_hbm_bytes_per_launch"] = out["kernels"][k[:100]]["hbm_bytes_per_launch"]
            out["pipe_hbm_bytes_per_step"] = out["pipe_hbm_bytes_per_launch"] / 9600.0
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r03_pmc_wavernn.json"), "w"), indent=1)
print({k: v for k, v in out.items() if k.startswith("pipe")})
