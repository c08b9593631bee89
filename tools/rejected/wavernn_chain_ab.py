"""A/B of the WaveRNN production chain on one GPU: classic (both GRU halves on the dependent path)
vs split-hidden (hidden halves beside fc1/fc2 of the previous step, rnn1 elementwise).
python tools/wavernn_chain_ab.py [frames]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import synth
from mockingbird_amd.vocoder.wavernn.inference import WaveRNNDevice

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
model = WaveRNNDevice(synth.wavernn_state(seed=5)["model_state"])
mel = torch.from_numpy(synth.wavernn_mel(frames, seed=1) / 4.0).cuda()
out, streams = [], {}
for chain in ("classic", "split", "split+merged"):
    os.environ["MBHIP_WAVERNN_CHAIN"] = chain.split("+")[0]
    os.environ["MBHIP_WAVERNN_MERGE"] = "1" if chain.endswith("merged") else "0"
    best = 1e9
    for rep in range(3):
        s = model.generate_samples(mel, True, 8000, 800, seed=7)
        torch.cuda.synchronize()
        best = min(best, model.last_loop_ms)
    streams[chain] = s.cpu()
    p = model.last_plan
    out.append(dict(chain=chain, loop_ms=best, us_per_step=best * 1e3 / p.seq_len, folds=p.n_folds, steps=p.seq_len))
    print(out[-1], flush=True)
a = streams["classic"]
for name in ("split", "split+merged"):
    agree = a == streams[name]
    first_bad = [int((~agree[i]).nonzero()[0]) if (~agree[i]).any() else a.shape[1] for i in range(a.shape[0])]
    print(name, "vs classic: first disagreement per fold (same Philox noise; S = none):", first_bad)
    out.append(dict(chain=name, first_disagreement=first_bad))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "wavernn_chain_ab.json"), "w"), indent=1)
