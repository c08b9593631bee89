"""bench.py's wavernn_batch32 object (32 utterances x mel 80x1000 = 736 fold columns x 9600 steps in ONE sample loop): the wide
recurrent GEMM on the fp16 matrix pipe (rnn_ts3_body.h, default) against the fp32 form (rnn_ts2_body.h, MBHIP_RNN_TS3=0).
usage: python tools/wrn_batch32_ab.py -> gpurun_out/wrn_batch32_ab.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, synth
from mockingbird_amd.vocoder.wavernn.inference import WaveRNNDevice
dev = WaveRNNDevice(synth.wavernn_state(seed=1)["model_state"])
mels = [torch.from_numpy(synth.wavernn_mel(1000, seed=100 + u) / 4.0).cuda() for u in range(32)]
seeds = list(range(500, 532))
out = {}
for mode in ("ts3", "ts2", "ts3"):
    os.environ["MBHIP_RNN_TS3"] = "0" if mode == "ts2" else "1"
    outs = dev.generate_samples_batch(mels, 8000, 800, seeds)
    torch.cuda.synchronize()
    us = dev.last_loop_ms * 1e3 / outs[0].shape[1]
    out.setdefault(mode, []).append(us)
    print(mode, "us per step", us, "columns", dev.last_batch_plan.n_folds, flush=True)
    if mode == "ts2":
        ref = outs[0].cpu()
    else:
        mine = outs[0].cpu()
    del outs
out["first_utterance_equal_fraction"] = float((ref == mine).float().mean())
print(json.dumps(out))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "wrn_batch32_ab.json"), "w"), indent=1)
