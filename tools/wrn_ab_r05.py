"""Round-5 A/B of the resident WaveRNN kernels: us per step of the default call (wavernn_pipe16.h) at BASELINE configs[1] and at wider
geometries, the exact kernel and the launch chain beside it, for whichever library MBHIP_LIB selects (run it once per library inside
ONE gpurun call: boxes of the pool differ by up to 10 % in clock).
usage: [MBHIP_LIB=build_variants/libmbhip_r04.so] python tools/wrn_ab_r05.py <tag> -> gpurun_out/wrn_ab_r05_<tag>.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch, synth
from mockingbird_amd.vocoder.wavernn.inference import WaveRNNDevice
tag = sys.argv[1] if len(sys.argv) > 1 else "x"
quick = len(sys.argv) > 2 and sys.argv[2] == "quick"
dev = WaveRNNDevice(synth.wavernn_state(seed=5)["model_state"])
out = {"lib": os.environ.get("MBHIP_LIB", "default"), "cases": {}}


def run(mel, target, overlap, reps, env=None):
    old = os.environ.get("MBHIP_WAVERNN_RESIDENT")
    if env is not None:
        os.environ["MBHIP_WAVERNN_RESIDENT"] = env
    try:
        us = []
        for r in range(reps):
            s = dev.generate_samples(mel, True, target, overlap, seed=5)
            torch.cuda.synchronize()
            us.append(dev.last_loop_ms * 1e3 / s.shape[1])
        return {"us_min": min(us), "us_median": float(np.median(us)), "columns": int(s.shape[0]), "steps": int(s.shape[1]),
                "launches": dev.last_loop_launches, "path": getattr(dev, "last_path", None), "fallback": getattr(dev, "last_fallback", None)}, s
    finally:
        if env is not None:
            if old is None:
                os.environ.pop("MBHIP_WAVERNN_RESIDENT", None)
            else:
                os.environ["MBHIP_WAVERNN_RESIDENT"] = old


mel = torch.from_numpy(synth.wavernn_mel(1000, seed=1) / 4.0).cuda()
run(mel[:, :60], 2000, 200, 1)  # warm-up
res, s_def = run(mel, 8000, 800, 7)
out["cases"]["configs1_default"] = res
print("configs1_default", json.dumps(res), flush=True)
res, s_x = run(mel, 8000, 800, 3, "exact")
res["identical_to_default"] = bool(torch.equal(s_x, s_def))
res["agree_fraction"] = float((s_x == s_def).float().mean())
out["cases"]["configs1_exact"] = res
print("configs1_exact", json.dumps(res), flush=True)
if not quick:
    res, s_c = run(mel, 8000, 800, 1, "0")
    res["identical_to_exact"] = bool(torch.equal(s_c, s_x))
    out["cases"]["configs1_chain"] = res
    print("configs1_chain", json.dumps(res), flush=True)
    for name, F, target, overlap in (("32_folds", 330, 2000, 100), ("42_folds", 330, 1500, 100), ("63_folds", 330, 1000, 50), ("88_folds", 330, 700, 50),
                                     ("69_folds_3000_frames", 3000, 8000, 800)):
        m = torch.from_numpy(synth.wavernn_mel(F, seed=2) / 4.0).cuda()
        res, _ = run(m, target, overlap, 3)
        out["cases"][name] = res
        print(name, json.dumps(res), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"wrn_ab_r05_{tag}.json"), "w"), indent=1)
