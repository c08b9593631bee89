"""Derive per-kernel MFMA utilisation of the HiFi-GAN fp16 pass from the counter passes of tools/pmc_gan.sh
(gpurun_out/pmc_gan_*.json) -> profiles/r01_hifigan_f16_mfma_util.json.

MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs * active cycles), SIMDs = 256 CUs * 4, active cycles =
GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs).  SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe
cycles per SIMD (32 per v_mfma_f32_32x32x16_f16, MI355X_MICROARCH.md).  Wave-state counters are quad-cycles and
include the four support waves of the fused kernel, which sit at barriers most of the time."""
import glob, json, os
RND = os.environ.get("MB_PMC_ROUND", "r01")  # output: profiles/<RND>_hifigan_<dtype>_mfma_util.json
DT = os.environ.get("MB_PMC_DTYPE", "f16")   # f32: the fp32-result path (conv1d_split_kernel, resblock_stage_f32_kernel)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = {}
for f in glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_gan_*.json")):
    for k, cs in json.load(open(f)).items():
        if any(n in k for n in ("resblock_pair", "resblock_stage", "conv1d_f16", "convt_f16", "conv_c1_f16", "conv_pw_f16", "cm_f32_to_tm_f16", "conv1d_split", "resblock_stage_f32", "conv1d_mfma", "conv_split_tm", "conv_c1_tm", "f32_transpose")):
            acc.setdefault(k, {}).update({c: v["mean_per_dispatch"] for c, v in cs.items()})
            acc[k]["dispatches"] = max(acc[k].get("dispatches", 0), max(v["dispatches"] for v in cs.values()))
out = {"source": "rocprofv3 --kernel-trace --pmc <group> (4 separate passes), python tools/gan_run.py hifigan %s 32 200 3; " % DT +
                 "means per dispatch", "kernels": {}}
tot_busy = tot_cyc = tot_valu = tot_mfma = 0.0
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0) * kv[1].get("dispatches", 0)):
    if "GRBM_GUI_ACTIVE" not in c or "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
        continue
    active = c["GRBM_GUI_ACTIVE"] / 8.0
    util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * active)
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    out["kernels"][k] = {
        "dispatches": c["dispatches"], "active_cycles": active, "mfma_util": util,
        "wave_wait_any_frac": c.get("SQ_WAIT_ANY", 0.0) / wc if wc else None,
        "wave_issue_stall_frac": c.get("SQ_WAIT_INST_ANY", 0.0) / wc if wc else None,
        "wave_active_frac": c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc if wc else None,
        "raw": {n: v for n, v in c.items() if n != "dispatches"},
    }
    tot_busy += c["SQ_VALU_MFMA_BUSY_CYCLES"] * c["dispatches"]
    tot_cyc += 1024.0 * active * c["dispatches"]
    if "SQ_INSTS_VALU" in c and "SQ_INSTS_MFMA" in c and c["SQ_INSTS_MFMA"]:
        # (SQ_INSTS_VALU counts the MFMAs too: the vector instructions BESIDE the MFMAs are the difference)
        out["kernels"][k]["valu_per_mfma"] = (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"]
        tot_valu += (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) * c["dispatches"]; tot_mfma += c["SQ_INSTS_MFMA"] * c["dispatches"]
out["whole_pass_mfma_util"] = tot_busy / tot_cyc if tot_cyc else None
out["whole_pass_valu_per_mfma"] = tot_valu / tot_mfma if tot_mfma else None
json.dump(out, open(os.path.join(ROOT, "profiles", RND + "_hifigan_" + DT + "_mfma_util.json"), "w"), indent=1)
for k, v in out["kernels"].items():
    print(f"{k[9:60]:52s} x{v['dispatches']:3d}  MfmaUtil {100 * v['mfma_util']:5.1f}%  wait {100 * (v['wave_wait_any_frac'] or 0):4.0f}%  issue-stall {100 * (v['wave_issue_stall_frac'] or 0):4.0f}%")
print("whole pass MfmaUtil %.1f%%" % (100 * out["whole_pass_mfma_util"]), " vector instructions beside each MFMA:", out["whole_pass_valu_per_mfma"])
