"""bench.py's wavernn_batch32 object (32 utterances x mel 80x1000 = 736 fold columns x 9600 steps in ONE sample loop): the wide
recurrent GEMM on the fp16 matrix pipe (rnn_ts3_body.h, default) against the fp32 form (rnn_ts2_body.h, MBHIP_RNN_WIDE=ts2).
usage: python tools/wrn_batch32_ab.py [modes, e.g. ts3,ts3nt6,ts2] -> gpurun_out/wrn_batch32_ab.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, synth
from mockingbird_amd.vocoder.wavernn.inference import WaveRNNDevice
dev = WaveRNNDevice(synth.wavernn_state(seed=1)["model_state"])
mels = [torch.from_numpy(synth.wavernn_mel(1000, seed=100 + u) / 4.0).cuda() for u in range(32)]
seeds = list(range(500, 532))
out = {}
modes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["ts3", "ts2", "ts3"]   # e.g. ts3,ts3nt6,ts2,ts3nt6,ts3
for mode in modes:
    os.environ["MBHIP_RNN_WIDE"] = ("ts2" if mode.startswith("ts2") else "ts3") + (":" + mode.split("nt")[1] if "nt" in mode else "")
    outs = dev.generate_samples_batch(mels, 8000, 800, seeds)
    torch.cuda.synchronize()
    us = dev.last_loop_ms * 1e3 / outs[0].shape[1]
    out.setdefault(mode, []).append(us)
    print(mode, "us per step", us, "columns", dev.last_batch_plan.n_folds, flush=True)
    if mode == "ts2":
        ref = outs[0].cpu()
    elif mode == modes[0]:
        mine = outs[0].cpu()
    else:
        out.setdefault(mode + "_equals_" + modes[0], []).append(float((outs[0].cpu() == mine).float().mean()))
    del outs
if "ts2" in modes: out["first_utterance_equal_fraction"] = float((ref == mine).float().mean())
print(json.dumps(out))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "wrn_batch32_ab.json"), "w"), indent=1)
