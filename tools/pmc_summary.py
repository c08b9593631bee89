"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: mean counter value per dispatch.
usage: pmc_summary.py <dir-with-*counter_collection.csv> <out.json>"""
import csv, glob, json, os, sys, collections
d, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name") or row.get("Kernel Name")
            c = row.get("Counter_Name") or row.get("Counter Name")
            v = float(row.get("Counter_Value") or row.get("Counter Value") or 0)
            a = acc[k][c]; a[0] += v; a[1] += 1
res = {k: {c: {"mean_per_dispatch": a[0] / max(a[1], 1), "dispatches": a[1]} for c, a in cs.items()} for k, cs in acc.items()}
json.dump(res, open(out, "w"), indent=1)
for k, cs in sorted(res.items(), key=lambda kv: -max(x["dispatches"] for x in kv[1].values()))[:12]:
    print(k[:90], {c: round(v["mean_per_dispatch"], 1) for c, v in cs.items()})
