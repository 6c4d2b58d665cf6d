#!/usr/bin/env python3
"""tools/kt_gaps.py <kernel_trace.csv> <name a> <name b> -- gaps between consecutive kernels whose names contain a / b (a rocprofv3
--kernel-trace of a bench run): end of a -> start of b, end of b -> start of a, durations."""
import csv, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
a, b = sys.argv[2], sys.argv[3]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if a in r["Kernel_Name"] or b in r["Kernel_Name"]]
out = {"a->b gap": [], "b->a gap": [], "a dur": [], "b dur": [], "a->a period": []}
last_a = None
for x, y in zip(sel[:-1], sel[1:]):
    g = int(y["Start_Timestamp"]) - int(x["End_Timestamp"])
    xa, ya = a in x["Kernel_Name"], a in y["Kernel_Name"]
    if xa and not ya: out["a->b gap"].append(g)
    if not xa and ya: out["b->a gap"].append(g)
    (out["a dur"] if xa else out["b dur"]).append(int(x["End_Timestamp"]) - int(x["Start_Timestamp"]))
    if xa:
        if last_a is not None: out["a->a period"].append(int(x["Start_Timestamp"]) - last_a)
        last_a = int(x["Start_Timestamp"])
for k, v in out.items():
    if v:
        v = np.array(v) / 1e3
        print(f"{k:14s} n {len(v):5d}  p10 {np.percentile(v, 10):8.2f}  p50 {np.percentile(v, 50):8.2f}  p90 {np.percentile(v, 90):8.2f} us")
