"""Sweep of wavernn_pipe.h's A/B switches at BASELINE configs[1] (23 folds): us per step per MBHIP_DIAG=wq_flags / wq_groups value.
usage: python tools/wrn_pipe_sweep.py FLAGS[,FLAGS...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from _diag import diag_set, diag_get
import torch, synth
from mockingbird_amd.vocoder.wavernn.inference import WaveRNNDevice
dev = WaveRNNDevice(synth.wavernn_state(seed=1)["model_state"])
mel = torch.from_numpy(synth.wavernn_mel(1000, seed=0) / 4.0).cuda()
_res = os.environ.get("MBHIP_WAVERNN_RESIDENT", "1")
os.environ["MBHIP_WAVERNN_RESIDENT"] = "0"
ref = dev.generate_samples(mel, True, 8000, 800, seed=5)
print("chain", dev.last_loop_ms * 1e3 / ref.shape[1], flush=True)
os.environ["MBHIP_WAVERNN_RESIDENT"] = _res  # "exact": the fp32 kernel
out = {}
for fl in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0"]):
    diag_set("wq_flags", fl)
    best = 1e9
    for rep in range(3):
        s = dev.generate_samples(mel, True, 8000, 800, seed=5)
        best = min(best, dev.last_loop_ms * 1e3 / s.shape[1])
    out[fl] = {"us_per_step": best, "identical": bool(torch.equal(s, ref)), "launches": dev.last_loop_launches}
    print(fl, out[fl], flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "wq_sweep.json"), "w"), indent=1)
