"""Round 5: profiles/r05_pmc_wavernn.json (HBM bytes per launch of the resident wf_pipe16_kernel: 2 x FETCH_SIZE + WRITE_SIZE of two separate
rocprofv3 --pmc passes over tools/wrn_run.py 1000 1 = BASELINE configs[1]) and profiles/r05_wavernn_pipe16_sq_counters.json (the two SQ
counter sets of tools/pmc_wavernn_sq.sh: how the kernel's waves spend their cycles, instructions per wave and step)."""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
try:
    f = json.load(open(os.path.join(G, "pmc4_wavernn_FETCH_SIZE.json")))
    w = json.load(open(os.path.join(G, "pmc4_wavernn_WRITE_SIZE.json")))
    out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two separate passes) -- python tools/wrn_run.py 1000 1 "
                     "(BASELINE configs[1]: mel 80x1000 -> 23 folds x 9600 steps, ONE resident launch of wf_pipe16_kernel); KB per dispatch; "
                     "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B)", "kernels": {}}
    for k in f:
        if k in w and "FETCH_SIZE" in f[k] and "WRITE_SIZE" in w[k]:
            fe, wr = f[k]["FETCH_SIZE"]["mean_per_dispatch"], w[k]["WRITE_SIZE"]["mean_per_dispatch"]
            out["kernels"][k[:100]] = {"FETCH_SIZE_KB": fe, "WRITE_SIZE_KB": wr, "dispatches": f[k]["FETCH_SIZE"]["dispatches"],
                                       "hbm_bytes_per_launch": (2.0 * fe + wr) * 1024.0}
            if "wf_pipe16_kernel" in k:
                out["pipe_hbm_bytes_per_launch"] = out["kernels"][k[:100]]["hbm_bytes_per_launch"]
                out["pipe_hbm_bytes_per_step"] = out["pipe_hbm_bytes_per_launch"] / 9600.0
    json.dump(out, open(os.path.join(ROOT, "profiles", "r05_pmc_wavernn.json"), "w"), indent=1)
    print({k: v for k, v in out.items() if k.startswith("pipe")})
except Exception as e:
    print("r05_pmc_wavernn.json skipped:", e)
try:
    cnt, name = {}, None
    for tag in ("a", "b"):
        d = json.load(open(os.path.join(G, f"pmc_wavernn_sq_{tag}.json")))
        for k, cs in d.items():
            if "wf_pipe16_kernel" in k:
                name = k
                for c, v in cs.items():
                    cnt[c] = v["mean_per_dispatch"]
    waves, steps = cnt["SQ_WAVES"], 9600.0
    kc = cnt["GRBM_GUI_ACTIVE"] / 8.0  # per XCD
    out = {"source": "tools/pmc_wavernn_sq.sh (rocprofv3 --kernel-trace --pmc, two counter sets, separate passes) over one launch of tools/wrn_run.py 1000 1 "
                     "(BASELINE configs[1]: 23 folds x 9600 steps, ONE launch of wf_pipe16_kernel, 224 workgroups x 8 waves)",
           "counters": cnt, "kernel": name,
           "derived": {"kernel_cycles": kc, "cycles_per_step": kc / steps,
                       "wave_wait_any_fraction": cnt["SQ_WAIT_ANY"] / cnt["SQ_WAVE_CYCLES"],
                       "wave_wait_inst_fraction": cnt["SQ_WAIT_INST_ANY"] / cnt["SQ_WAVE_CYCLES"],
                       "wave_active_fraction": cnt["SQ_ACTIVE_INST_ANY"] / cnt["SQ_WAVE_CYCLES"],
                       "mfma_busy_fraction_of_all_simds": cnt["SQ_VALU_MFMA_BUSY_CYCLES"] / (cnt["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4),
                       "per_wave_per_step": {k2: cnt[c] / waves / steps for k2, c in (("mfma", "SQ_INSTS_MFMA"), ("valu", "SQ_INSTS_VALU"), ("salu", "SQ_INSTS_SALU"),
                                                                                     ("vmem_rd", "SQ_INSTS_VMEM_RD"), ("vmem_wr", "SQ_INSTS_VMEM_WR"), ("lds", "SQ_INSTS_LDS"))}}}
    out["derived"]["per_wave_per_step"]["valu_plus_salu"] = out["derived"]["per_wave_per_step"]["valu"] + out["derived"]["per_wave_per_step"]["salu"]
    json.dump(out, open(os.path.join(ROOT, "profiles", "r05_wavernn_pipe16_sq_counters.json"), "w"), indent=1)
    print(out["derived"])
except Exception as e:
    print("r05_wavernn_pipe16_sq_counters.json skipped:", e)
