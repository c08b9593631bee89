"""A/B of the ppg2mel decoder loop for a batch: ONE resident launch (csrc/ppg_batch.h, 2..32 utterances) against the 6-launch graph-replayed
step (csrc/ppg_fast.h).  Injected masks first (results must agree to fp32 rounding and stop at the same step), then the device RNG for timing.
usage: python tools/ppg_batch_ab.py [B,B,...] [T] -> one JSON line per batch size (gpurun_out/ppg_batch_ab.json)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, synth, hiputil
from mockingbird_amd.ppg2mel import Ppg2MelDecoder
Bs = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2, 5, 16, 17, 32]
T = int(sys.argv[2]) if len(sys.argv) > 2 else 40
out = {}
for B in Bs:
    dec = Ppg2MelDecoder(synth.ppg2mel_decoder_state(synth.PPG2MEL_HP, seed=3, stop_bias=0.0 if T < 100 else -8.0), synth.PPG2MEL_HP)
    mem = torch.from_numpy(synth.ppg2mel_memory(B, T, seed=1 + B)).cuda()
    masks = synth.ppg2mel_dropout_masks(4, T * 2, B)
    res, info = {}, {}
    for mode in ("0", "1"):
        os.environ["MBHIP_PPG_RESIDENT"] = mode
        t0 = time.perf_counter()
        m, a, s = dec.decode(mem, dropout=masks)
        torch.cuda.synchronize()
        res[mode] = (m.cpu(), a.cpu(), s.cpu())
        info[mode] = {"steps": int(a.shape[1]), "launches": dec.last_loop_launches, "us_per_step": dec.last_loop_ms * 1e3 / max(1, a.shape[1]),
                      "wall_ms": (time.perf_counter() - t0) * 1e3}
    same = res["0"][1].shape == res["1"][1].shape
    n = min(res["0"][1].shape[1], res["1"][1].shape[1])
    r = {"chain": info["0"], "resident": info["1"], "same_steps": same,
         "mel_max_abs_diff": hiputil.relerr(res["1"][0].reshape(B, -1, 80)[:, :2 * n], res["0"][0].reshape(B, -1, 80)[:, :2 * n])["max_abs"] if n else None,
         "align_max_abs_diff": float((res["1"][1][:, :n] - res["0"][1][:, :n]).abs().max()) if n else None,
         "stop_max_abs_diff": float((res["1"][2][:, :n] - res["0"][2][:, :n]).abs().max()) if n else None,
         "finite": bool(torch.isfinite(res["1"][0]).all())}
    # timing with the device RNG, no stop (max_steps forced)
    t_us = {}
    for mode in ("0", "1"):
        os.environ["MBHIP_PPG_RESIDENT"] = mode
        us = []
        for i in range(3):
            m, a, s = dec.decode(mem, seed=2, stop_threshold=2.0)
            torch.cuda.synchronize()
            us.append(dec.last_loop_ms * 1e3 / max(1, a.shape[1]))
        t_us["chain" if mode == "0" else "resident"] = {"us_per_step": sorted(us)[1], "steps": int(a.shape[1]), "launches": dec.last_loop_launches}
    r["device_rng_never_stopping"] = t_us
    out[f"B{B}"] = r
    print(f"B{B}", json.dumps(r), flush=True)
os.environ.pop("MBHIP_PPG_RESIDENT", None)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "ppg_batch_ab.json"), "w"), indent=1)
