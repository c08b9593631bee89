"""A/B sweep of the WaveRNN sample-loop launch structure on one GPU (lanes x graph on/off)."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import synth
from mockingbird_amd.vocoder.wavernn.inference import WaveRNNDevice

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 400
model = WaveRNNDevice(synth.wavernn_state(seed=5)["model_state"])
mel = torch.from_numpy(synth.wavernn_mel(frames, seed=1) / 4.0).cuda()
out = []
for graph in (1,):
    for lanes in (1,):
        os.environ["MBHIP_WAVERNN_LANES"] = str(lanes)
        if graph:
            os.environ.pop("MBHIP_NO_GRAPH", None)
        else:
            os.environ["MBHIP_NO_GRAPH"] = "1"
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            s = model.generate_samples(mel, True, 8000, 800, seed=rep)
            wall = time.perf_counter() - t0
            best = min(best, model.last_loop_ms)
        p = model.last_plan
        out.append(dict(graph=graph, lanes=lanes, loop_ms=best, us_per_step=best * 1e3 / p.seq_len, folds=p.n_folds,
                        wall_ms=wall * 1e3))
        print(out[-1], flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "wavernn_sweep.json"), "w"), indent=1)
