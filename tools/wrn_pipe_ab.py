"""A/B of the WaveRNN sample loop at BASELINE configs[1] (23 fold columns) and neighbours: 5-launch chain (wavernn_fast.h,
hipGraph replays) vs ONE resident launch with role-specialised workgroups and two column groups in flight
(wavernn_pipe.h).  Same seed -> the sample streams must be identical.
usage: python tools/wrn_pipe_ab.py [out.json] -> gpurun_out/wavernn_pipe_ab.json (+ the kernel's wall-clock marks)"""
import json, os, struct, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from _diag import diag_set, diag_get
import torch, synth
from mockingbird_amd.vocoder.wavernn.inference import WaveRNNDevice
dev = WaveRNNDevice(synth.wavernn_state(seed=1)["model_state"])
out = {"cases": {}}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
CASES = (("configs1_23_folds", 1000, 8000, 800, ("chain", "pipe")), ("12_folds", 200, 3000, 300, ("chain", "pipe", "pipe_1group")),
         ("4_folds", 40, 2200, 100, ("chain", "pipe", "persist")), ("32_folds", 330, 2000, 100, ("chain", "pipe")))
if os.environ.get("WQ_AB_CASES"):  # e.g. WQ_AB_CASES=configs1_23_folds: only these cases (a quick marks run under MBHIP_DIAG=wq_flags=<bits>)
    CASES = tuple(c for c in CASES if c[0] in os.environ["WQ_AB_CASES"].split(","))
for name, F, target, overlap, modes in CASES:
    mel = torch.from_numpy(synth.wavernn_mel(F, seed=0) / 4.0).cuda()
    res = {}
    for mode in modes:
        diag_set("wq_groups")
        diag_set("wp_trace")
        # (WQ_AB_EXACT=1: the exact fp32 resident kernel instead of the operand-pair one; "persist": the 4-column persist kernel of
        #  round 3 is gone, the mode now times the default resident kernel)
        os.environ["MBHIP_WAVERNN_RESIDENT"] = ("exact" if os.environ.get("WQ_AB_EXACT") == "1" else "1") if mode != "chain" else "0"
        if mode == "pipe_1group":
            diag_set("wq_groups", "1")
        dev.generate_samples(mel[:, :40], True, 2200, 100, seed=2)  # warm-up
        best = None
        for rep in range(3):
            smp = dev.generate_samples(mel, True, target, overlap, seed=5)
            torch.cuda.synchronize()
            us = dev.last_loop_ms * 1e3 / smp.shape[1]
            best = us if best is None else min(best, us)
        res[mode] = {"us_per_step": best, "launches": dev.last_loop_launches, "columns": int(smp.shape[0]), "steps": int(smp.shape[1])}
        res[mode + "_samples"] = smp
        if mode == "pipe" and smp.shape[1] > 1100:  # wall-clock marks (100 MHz) of one workgroup per role, group 0, steps 1000..1003
            tf = os.path.join(ROOT, "gpurun_out", f"wq_trace_{name}.bin")
            diag_set("wp_trace", tf)
            dev.generate_samples(mel, True, target, overlap, seed=5)
            diag_set("wp_trace")
            if os.path.exists(tf):
                raw = open(tf, "rb").read()
                m = struct.unpack("<320Q", raw[:2560])
                if len(raw) >= 8192:  # round 5: every workgroup's publish time at step 1001 (R1 0..63, R2 64..127, F1 / F2 / F3 32 each)
                    pw = struct.unpack("<224Q", raw[4096:4096 + 224 * 8])
                    res["publish_us_step1001"] = {role: [round((x - m[0]) / 100.0, 2) if x else None for x in pw[lo:hi]]
                                                  for role, lo, hi in (("R1", 0, 64), ("R2", 64, 128), ("F1", 128, 160), ("F2", 160, 192), ("F3", 192, 224))}
                t0 = m[0]  # R1, step 1000, mark 0
                res["pipe_marks_us"] = {role: [[(m[(r * 4 + st) * 16 + k] - t0) / 100.0 if m[(r * 4 + st) * 16 + k] else None for k in range(9)]
                                               for st in range(4)] for r, role in enumerate(("R1", "R2", "F1", "F2", "F3"))}
    ref = res.pop("chain_samples")
    res["sample_streams_identical"] = {m: bool(torch.equal(ref, res.pop(m + "_samples"))) for m in modes[1:]}
    res["speedup_pipe"] = res["chain"]["us_per_step"] / res["pipe"]["us_per_step"]
    out["cases"][name] = res
    print(name, json.dumps(res), flush=True)
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "wavernn_pipe_ab.json"), "w"), indent=1)
