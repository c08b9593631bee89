"""Where does run-to-run nondeterminism of TacotronDevice.generate (B=32) come from?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import synth
from mockingbird_amd.synthesizer.inference import TacotronDevice

st = synth.tacotron_state(seed=3)["model_state"]
dev = TacotronDevice(st, torch.device("cuda"))
seqs, emb = synth.tacotron_inputs(32, 90, 110, seed=2)
T = max(len(s) for s in seqs)
chars = torch.tensor(np.stack([np.pad(s, (0, T - len(s))) for s in seqs])).long().cuda()
spk = torch.tensor(np.stack(emb)).cuda()

def d(a, b):
    return float((a - b).abs().max())

for style in (-1, 0):
    enc = [dev.encode(chars, spk, style, None, 5) for _ in range(3)]
    print("style", style, "encode mem diffs", [d(enc[0][0], e[0]) for e in enc[1:]], "proj", [d(enc[0][1], e[1]) for e in enc[1:]])
    from mockingbird_amd.synthesizer import frontend
    from mockingbird_amd.synthesizer.hparams import hparams
    sty = [frontend.style_embed(dev.front, hparams, spk, style) for _ in range(3)]
    print("   style_embed diffs", [d(sty[0], s) for s in sty[1:]])
mem, memp = enc[0]
mem = mem.clone(); memp = memp.clone()
for B in (8, 16, 17, 32):
    for steps in (2, 8, 40, 400):
        outs = [dev.decode(mem[:B].contiguous(), memp[:B].contiguous(), chars[:B].contiguous(), steps, 11.0, None, 5) for _ in range(3)]
        outs = [(m.clone(), l.clone(), a.clone()) for m, l, a in outs]
        print(f"B={B} steps={steps} mel", [d(outs[0][0], o[0]) for o in outs[1:]], "lin", [d(outs[0][1], o[1]) for o in outs[1:]],
              "attn", [d(outs[0][2], o[2]) for o in outs[1:]])
        if steps == 400:
            m0, m1 = outs[0][0], outs[1][0]
            bad = ((m0 - m1).abs() > 0).any(dim=1)  # [B][frames]
            if bad.any():
                fr = bad.any(dim=0).nonzero().flatten()
                print("   first differing frame", int(fr[0]), "items", bad[:, int(fr[0])].nonzero().flatten().tolist()[:8])
