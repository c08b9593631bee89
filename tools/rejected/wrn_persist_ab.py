"""A/B of the WaveRNN sample loop for few fold columns: 5-launch chain (wavernn_fast.h, hipGraph replays) vs ONE persistent
launch with resident weights (wavernn_persist.h).  Same seed -> the sample streams must be identical.
usage: python tools/wrn_persist_ab.py [frames] -> gpurun_out/wavernn_persistent_ab.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, synth
from mockingbird_amd.vocoder.wavernn.inference import WaveRNNDevice
F = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = WaveRNNDevice(synth.wavernn_state(seed=1)["model_state"])
mel = torch.from_numpy(synth.wavernn_mel(F, seed=0) / 4.0).cuda()
out = {"workload": f"mel 80x{F}", "cases": {}}
for name, batched, target, overlap in (("unbatched_1_column", False, 0, 0), ("batched_3_columns", True, 8000, 800)):
    res = {}
    modes = ("chain", "persistent", "persistent_mfma") if not batched else ("chain", "persistent")
    for mode in modes:
        os.environ["MBHIP_WAVERNN_PERSIST"] = "0" if mode == "chain" else "1"
        os.environ.pop("MBHIP_WP_MFMA", None)
        if mode == "persistent_mfma":
            os.environ["MBHIP_WP_MFMA"] = "1"
        dev.generate_samples(mel[:, :12], batched, 600, 50, seed=2)  # warm-up: graph capture / attribute / first touch
        best = None
        for rep in range(3):
            smp = dev.generate_samples(mel, batched, target, overlap, seed=5)
            torch.cuda.synchronize()
            us = dev.last_loop_ms * 1e3 / smp.shape[1]
            best = us if best is None else min(best, us)
        res[mode] = {"us_per_step": best, "launches": dev.last_loop_launches, "columns": int(smp.shape[0]), "steps": int(smp.shape[1])}
        res[mode + "_samples"] = smp
    ref = res.pop("chain_samples")
    res["sample_streams_identical"] = all(bool(torch.equal(ref, res.pop(m + "_samples"))) for m in modes[1:])
    res["speedup"] = res["chain"]["us_per_step"] / res["persistent"]["us_per_step"]
    out["cases"][name] = res
    print(name, json.dumps(res))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "wavernn_persistent_ab.json"), "w"), indent=1)
