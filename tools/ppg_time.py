"""ppg2mel decoder timing: fast step vs general loop, batch 1 and 32 (T_enc = 200, 400 steps forced)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, synth
from mockingbird_amd.ppg2mel import Ppg2MelDecoder
pdec = Ppg2MelDecoder(synth.ppg2mel_decoder_state(synth.PPG2MEL_HP, seed=3, stop_bias=-6.0), synth.PPG2MEL_HP)
out = {}
for name, env in (("fast", {}), ("general", {"MBHIP_PPG_FAST": "0"})):
    os.environ.pop("MBHIP_PPG_FAST", None); os.environ.update(env)
    for pb in (1, 32):
        mem = torch.from_numpy(synth.ppg2mel_memory(pb, 200, seed=1)).cuda()
        pdec.decode(mem, seed=1)
        ts = []
        for i in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            pm, _, _ = pdec.decode(mem, seed=2 + i)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        t = sorted(ts)[2]
        e = {"wall_us_per_step": t * 1e6 / pm.shape[1], "steps": int(pm.shape[1])}
        if name == "fast": e["loop_us_per_step"] = pdec.last_loop_ms * 1e3 / pdec.last_loop_steps
        out[f"{name}_b{pb}"] = e
        print(name, pb, e, flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "ppg_time.json"), "w"), indent=1)
