"""Tacotron decoder loop: fused front (taco_front_kernel, 5 launches per iteration) against one launch each (7), same box, and the
fused loop's diagnostics knobs.  Usage: python tools/taco_front_ab.py [out.json] [quick]"""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import synth
from oracle import tacotron as ot
from mockingbird_amd.synthesizer.inference import TacotronDevice

st = synth.tacotron_state(seed=3)["model_state"]
dev = TacotronDevice(st, torch.device("cuda"))
out = {}
VARIANTS = [("default", ""), ("no_gru_watch", "taco_gru_watch=0"), ("dma_late", "taco_dma_early=0"), ("five_f16", "taco_f16=1,taco_fold=0"), ("five_f32", "taco_f16=0"),
            ("seven", "taco_front=0"), ("fold_m1_160", "taco_m1=160"), ("fold_m1_229", "taco_m1=229"), ("default_again", "")]
SHAPES = ((32, 60, 100), (16, 60, 100), (1, 60, 60), (32, 150, 180))
if "quick" in sys.argv[2:]:
    SHAPES = SHAPES[:1]
for B, tmin, tmax in SHAPES:
    seqs, emb = synth.tacotron_inputs(B, tmin, tmax, seed=9)
    T = max(len(s) for s in seqs)
    chars = torch.tensor(np.stack([np.pad(s, (0, T - len(s))) for s in seqs])).long()
    with torch.no_grad():
        mem, memp = ot.encoder_memory(st, ot.HP, chars, torch.tensor(np.stack(emb)), 0)
    mem, memp, chars = mem.cuda(), memp.cuda(), chars.cuda()
    row = {}
    for name, diag in VARIANTS:
        os.environ["MBHIP_DIAG"] = diag
        ts = []
        for rep in range(4):
            dev.decode(mem, memp, chars, 800, 11.0, seed=5)
            ts.append(dev.last_loop_ms * 1e3 / dev.last_loop_iterations)
        row[name] = {"us_per_iteration": round(float(np.median(ts[1:])), 3), "launches": dev.last_loop_launches_per_iteration}
    out[f"B{B}_T{T}"] = row
    print(B, T, {k: v["us_per_iteration"] for k, v in row.items()}, flush=True)
os.environ["MBHIP_DIAG"] = ""
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
