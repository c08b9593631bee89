"""Phase timeline of one decoder iteration of the fast loop (taco_fast.h tf_mark stamps, MBHIP_DIAG=taco_trace=<file>).
Per kernel: shader-clock deltas between marks of ONE workgroup (cycles and us at the measured clock), and the
100 MHz wall clock of kernel start / end to see the gaps between the launches of the last iteration."""
import json, os, struct, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from _diag import diag_set, diag_get
out_path = os.path.join(ROOT, "gpurun_out", "taco_trace.bin")
diag_set("taco_trace", out_path)
SEVEN = "seven" in sys.argv[1:]  # one launch each for fc2 / GRU / attention instead of taco_front_kernel
if SEVEN:
    diag_set("taco_front", "0")
FIVE = "five" in sys.argv[1:]   # the fused front without the folded rnn_input (5 launches)
if FIVE:
    diag_set("taco_fold", "0")
import numpy as np, torch, synth
from mockingbird_amd.synthesizer.inference import TacotronDevice
st = synth.tacotron_state(seed=3)["model_state"]
dev = TacotronDevice(st, torch.device("cuda"))
seqs, emb = synth.tacotron_inputs(32, 90, 110, seed=2)
T = max(len(s) for s in seqs)
chars = torch.tensor(np.stack([np.pad(s, (0, T - len(s))) for s in seqs])).long().cuda()
spk = torch.tensor(np.stack(emb)).cuda()
mem, memp = dev.encode(chars, spk, -1, None, 1)
for _ in range(3):
    dev.decode(mem, memp, chars, 400, 11, seed=1)
torch.cuda.synchronize()
flat = np.fromfile(out_path, dtype=np.uint64).astype(np.int64)
raw = flat[:16 * 9].reshape(-1, 16)
wg = flat[16 * 9:].reshape(-1, 2) if flat.size > 16 * 9 else None
names = ["fc2", "gru", "lsa", "rin", "lstm1", "lstm2", "mel", "mel_fc1", "mel_stop"]
labels = {"lsa": ["start", "loads issued", "B1 (query+cum staged)", "B2 (pq)", "-", "B3 (energies)", "B4 (softmax)", "B5 (context partials)", "end"],
          "*": ["start", "loads issued", "wave0 MFMAs done", "reduction barrier", "end"]}
res = {}
wall0 = min(int(r[14]) for r in raw if r[14] > 0)
for name, r in zip(names, raw):
    if r[14] == 0:
        continue  # a launch this form does not have
    lab = labels.get(name, labels["*"])
    marks = [(lab[k], int(r[k])) for k in range(len(lab)) if lab[k] != "-" and r[k] > 0]
    wall = (int(r[14]) - wall0, int(r[15]) - wall0)  # 10 ns ticks
    cyc = marks[-1][1] - marks[0][1]
    us_wall = (wall[1] - wall[0]) * 0.01
    ghz = cyc / (us_wall * 1e3) if us_wall > 0 else 0
    res[name] = {"wall_start_us": wall[0] * 0.01, "wall_end_us": wall[1] * 0.01, "cycles": cyc, "clock_GHz": ghz,
                 "phases_cycles": {marks[i + 1][0]: marks[i + 1][1] - marks[i][1] for i in range(len(marks) - 1)}}
    print(f"{name:9s} wall {wall[0]*0.01:8.2f} -> {wall[1]*0.01:8.2f} us ({us_wall:5.2f} us, {cyc} cyc, {ghz:.2f} GHz)  " +
          "  ".join(f"{k}: {v}" for k, v in res[name]["phases_cycles"].items()))
res["rider_marks_us"] = {"rin_last_rider": (int(raw[3][13]) - wall0) * 0.01, "rin_last_pre_tile": (int(raw[3][12]) - wall0) * 0.01,
                         "rin_stop_tile": (int(raw[3][11]) - wall0) * 0.01, "mel_last_rider": (int(raw[6][13]) - wall0) * 0.01}
print(res["rider_marks_us"])
if not SEVEN and wg is not None:
    q = [(int(x[0]) - wall0) * 0.01 for x in wg if x[0] > 0]; e = [(int(x[1]) - wall0) * 0.01 for x in wg if x[1] > 0]
    if q and e:
        res["attention_workgroups_us"] = {"query_arrival_min_max": [min(q), max(q)], "end_min_max": [min(e), max(e)],
                                          "end_by_group_of_32": [round(float(np.mean(e[i:i + 32])), 2) for i in range(0, len(e), 32)],
                                          "latest_8": sorted(range(len(e)), key=lambda i: -e[i])[:8]}
        print(res["attention_workgroups_us"])
if not SEVEN:
    res["front_extra_us"] = {"last_attention_workgroup": (int(raw[2][13]) - wall0) * 0.01, "first_hh2_tile_done": (int(raw[0][12]) - wall0) * 0.01,
                             "last_hh2_tile_done": (int(raw[0][13]) - wall0) * 0.01}
    print(res["front_extra_us"])
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "taco_trace_seven.json" if SEVEN else "taco_trace_five.json" if FIVE else "taco_trace.json"), "w"), indent=1)
