"""(WaveRNN: tools/pmc_wavernn_r02.sh -- its fast chain has to be profiled one launch type at a time.)
Combine the FETCH_SIZE / WRITE_SIZE summaries of tools/pmc_r02.sh into profiles/r02_pmc_<what>.json: HBM bytes per
launch per kernel = 2 x FETCH_SIZE (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md) + WRITE_SIZE, both in
KB per dispatch, mean over all dispatches of the benchmarked configuration."""
import json, os, sys
RND = os.environ.get("MB_PMC_ROUND", "r02")  # output prefix: profiles/<RND>_pmc_<what>.json
ONLY = sys.argv[1:]                           # restrict to these objects (default: all that have summaries)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two separate passes), %s, MBHIP_NO_GRAPH=1 (counter mode "
       "crashes on hipGraph replays; same kernels and arguments); KB per dispatch, mean over all dispatches; FETCH_SIZE "
       "doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B)")
WHAT = {
    "tacotron": ("tools/taco_run.py (BASELINE configs[2]: B=32, ~100 tokens, 400 decoder iterations)",
                 {"front": ["taco_front_kernel"],  # round 5: fc2 | attention GRU | attention as roles of one launch
                  "prenet_fc2": ["taco_fc2_kernel"], "attn_gru": ["taco_gru_kernel"], "lsa": ["lsa_hh_kernel"],
                  "rnn_input": ["taco_rin_kernel"], "lstm": ["taco_lstm_kernel"], "mel_proj": ["taco_mel_kernel"]}),
    "hifigan": ("tools/gan_run.py hifigan f16 32 200 (bench object hifigan_f16)",
                {"resblock_pair": ["resblock_pair"], "resblock_stage": ["resblock_stage"], "conv1d_f16": ["conv1d_f16"]}),
    "hifigan_f32": ("tools/gan_run.py hifigan f32 32 200 (bench object hifigan: fp32 storage, error-compensated fp16 MFMA)",
                    {"resblock_stage_f32": ["resblock_stage_f32"], "conv1d_split": ["conv1d_split"], "conv1d_mfma": ["conv1d_mfma"],
                     "resblock_pair_split": ["resblock_pair_split"], "conv_split_tm": ["conv_split_tm"], "conv_c1_tm": ["conv_c1_tm"], "f32_transpose": ["f32_transpose"]}),  # round 6: the time-major plan
    "fregan": ("tools/gan_run.py fregan f16 8 3000 (bench object fregan_f16)",
               {"resblock_pair": ["resblock_pair"], "resblock_stage": ["resblock_stage"], "conv1d_f16": ["conv1d_f16"],
                "add_inplace": ["add_inplace_f16"]}),
}
for what, (cmd, names) in WHAT.items():
    if ONLY and what not in ONLY:
        continue
    try:
        f = json.load(open(os.path.join(ROOT, "gpurun_out", f"pmc2_{what}_FETCH_SIZE.json")))
        w = json.load(open(os.path.join(ROOT, "gpurun_out", f"pmc2_{what}_WRITE_SIZE.json")))
    except Exception as e:
        print(what, "skipped:", e)
        continue
    out = {"source": SRC % cmd, "kernels": {}}
    if what in ("hifigan", "hifigan_f32", "fregan"):
        out["forwards"] = 3  # tools/gan_run.py: one warm-up + two timed forwards
    for name, subs in names.items():
        # every instance (template arguments differ per layer shape): aggregate all kernels matching the substrings
        ks = [k for k in f if all(s in k for s in subs) and k in w]
        per = []
        for k in ks:
            fe, wr = f[k]["FETCH_SIZE"]["mean_per_dispatch"], w[k]["WRITE_SIZE"]["mean_per_dispatch"]
            per.append({"kernel": k[:160], "dispatches": f[k]["FETCH_SIZE"]["dispatches"], "FETCH_SIZE_KB": fe, "WRITE_SIZE_KB": wr,
                        "hbm_bytes_per_launch": (2.0 * fe + wr) * 1024.0})
        if not per:
            continue
        n = sum(p["dispatches"] for p in per)
        out["kernels"][name] = per
        out[name + "_hbm_bytes_per_launch"] = sum(p["hbm_bytes_per_launch"] * p["dispatches"] for p in per) / max(n, 1)
        out[name + "_hbm_bytes_total"] = sum(p["hbm_bytes_per_launch"] * p["dispatches"] for p in per)
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{RND}_pmc_{what}.json"), "w"), indent=1)
    print(what, json.dumps({k: round(v) for k, v in out.items() if k.endswith("per_launch")}))
