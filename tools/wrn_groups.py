"""Step time of the resident WaveRNN kernel (wavernn_pipe16.h) by fold-column count and column groups, beside the launch chain.
usage: python tools/wrn_groups.py -> gpurun_out/wrn_groups.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from _diag import diag_set
import torch, synth
from mockingbird_amd.vocoder.wavernn.inference import WaveRNNDevice
dev = WaveRNNDevice(synth.wavernn_state(seed=1)["model_state"])
out = {}
for name, F, target, overlap, groups in (("23_folds", 1000, 8000, 800, (2, 3, 4)), ("32_folds", 330, 2000, 100, (2, 4)), ("42_folds", 330, 1500, 100, (3, 4)),
                                         ("63_folds", 330, 1000, 50, (4,))):
    mel = torch.from_numpy(synth.wavernn_mel(F, seed=0) / 4.0).cuda()
    res = {}
    os.environ["MBHIP_WAVERNN_RESIDENT"] = "0"
    dev.generate_samples(mel, True, target, overlap, seed=5)
    s = dev.generate_samples(mel, True, target, overlap, seed=5)
    res["chain"] = {"us_per_step": dev.last_loop_ms * 1e3 / s.shape[1], "columns": int(s.shape[0]), "steps": int(s.shape[1])}
    os.environ["MBHIP_WAVERNN_RESIDENT"] = "1"
    for g in groups:
        diag_set("wq_groups", str(g))
        best = 1e9
        for rep in range(3):
            s = dev.generate_samples(mel, True, target, overlap, seed=5)
            best = min(best, dev.last_loop_ms * 1e3 / s.shape[1])
        res[f"pipe16_{g}_groups"] = {"us_per_step": best, "launches": dev.last_loop_launches}
    diag_set("wq_groups")
    out[name] = res
    print(name, json.dumps(res), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "wrn_groups.json"), "w"), indent=1)
