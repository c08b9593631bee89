"""A/B of the ppg2mel decoder loop at one utterance: resident launch (csrc/ppg_resident.h) against the 6-launch
graph-replayed step (csrc/ppg_fast.h) on bench.py's ppg2mel object (T_enc = 200, 400 steps forced, device RNG).
Prints one JSON line; with MBHIP_DIAG=pr_trace=<file> set, also the per-role wall-clock marks of steps 100..103."""
import json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _diag import diag_set, diag_get
import os
import struct
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import synth  # noqa: E402
import hiputil  # noqa: E402
from mockingbird_amd.ppg2mel import Ppg2MelDecoder  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dec = Ppg2MelDecoder(synth.ppg2mel_decoder_state(synth.PPG2MEL_HP, seed=3, stop_bias=-6.0), synth.PPG2MEL_HP)
mem = torch.from_numpy(synth.ppg2mel_memory(1, T, seed=1)).cuda()
out = {"T_enc": T}
res = {}
for mode in ("0", "1"):
    os.environ["MBHIP_PPG_RESIDENT"] = mode
    dec.decode(mem, seed=1)
    torch.cuda.synchronize()
    us, wall = [], []
    for i in range(5):
        t0 = time.perf_counter()
        m, a, s = dec.decode(mem, seed=2)
        torch.cuda.synchronize()
        wall.append((time.perf_counter() - t0) * 1e6 / m.shape[1])
        us.append(dec.last_loop_ms * 1e3 / m.shape[1])
    res[mode] = (m.cpu(), a.cpu(), s.cpu())
    out["resident" if mode == "1" else "chain"] = {"us_per_step": sorted(us)[2], "wall_us_per_step": sorted(wall)[2], "steps": int(m.shape[1]),
                                                   "launches": dec.last_loop_launches}
out["mel_max_abs_diff"] = hiputil.relerr(res["1"][0], res["0"][0])["max_abs"]
out["align_max_abs_diff"] = float((res["1"][1] - res["0"][1]).abs().max())
print(json.dumps(out))
tr = diag_get("pr_trace")
if tr and os.path.exists(tr):
    raw = open(tr, "rb").read()
    marks = struct.unpack("<%dQ" % (len(raw) // 8), raw)
    names = ["ATT", "DEC", "Q0", "OUT", "MOL"]
    base = min(x for x in marks if x)
    for r in range(5):
        for st in range(4):
            row = marks[(r * 4 + st) * 16:(r * 4 + st) * 16 + 16]
            print(names[r], 100 + st, " ".join("%7.2f" % ((x - base) / 100.0) if x else "      -" for x in row[:8]))
    if len(marks) >= 1000:
        for name, lo, hi in (("ATT publish", 512, 576), ("DEC publish", 576, 704), ("Q0 publish", 704, 712), ("OUT publish", 712, 728), ("ATT p0 fetched", 768, 832)):
            v = [(x - base) / 100.0 for x in marks[lo:hi] if x]
            if v:
                srt = sorted(range(len(v)), key=lambda i: v[i])
                print(name, "min %.2f max %.2f" % (min(v), max(v)), "latest:", [(i, round(v[i], 2)) for i in srt[-4:]], "earliest:", [(i, round(v[i], 2)) for i in srt[:2]])
