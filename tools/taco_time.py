"""Tacotron configs[2] decode timing (B=32, T~100, 200 iterations forced): fast loop vs general loop, per call
and per decoder iteration (postnet timed separately by decoding 2 frames)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch, synth
from mockingbird_amd.synthesizer.inference import TacotronDevice
st = synth.tacotron_state(seed=3)["model_state"]
dev = TacotronDevice(st, torch.device("cuda"))
seqs, emb = synth.tacotron_inputs(32, 90, 110, seed=2)
T = max(len(s) for s in seqs)
chars = torch.tensor(np.stack([np.pad(s, (0, T - len(s))) for s in seqs])).long().cuda()
spk = torch.tensor(np.stack(emb)).cuda()
mem, memp = dev.encode(chars, spk, -1, None, 1)
out = {}
for name, env in (("fast", {}), ("fast_nograph", {"MBHIP_NO_GRAPH": "1"}), ("general", {"MBHIP_TACO_FAST": "0"})):
    for k in ("MBHIP_NO_GRAPH", "MBHIP_TACO_FAST"):
        os.environ.pop(k, None)
    os.environ.update(env)
    res = {}
    for steps in (400, 40):
        dev.decode(mem, memp, chars, steps, 11, seed=1)
        ts = []
        for i in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            dev.decode(mem, memp, chars, steps, 11, seed=1)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        res[steps] = sorted(ts)[2]
    per_iter = (res[400] - res[40]) / 180 * 1e6
    out[name] = {"decode400_ms": res[400] * 1e3, "decode40_ms": res[40] * 1e3, "us_per_iteration_incl_postnet_growth": per_iter}
    if name != "general":  # HIP events around the loop itself (mb_taco_last_loop_ms)
        dev.decode(mem, memp, chars, 400, 11, seed=1)
        out[name]["loop_ms"] = dev.last_loop_ms
        out[name]["loop_us_per_iteration"] = dev.last_loop_ms * 1e3 / dev.last_loop_iterations
    print(name, out[name], flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "taco_time.json"), "w"), indent=1)
