#!/usr/bin/env python3
"""Summarises the split-bf16 FIR kernel's per-wave phase clocks (RSMP_FIR_WTRACE=<file>): mean shader
cycles per item spent in each phase, per wave of the workgroup (averaged over workgroups)."""
import sys
import numpy as np
NAMES = {1: "wait staged", 2: "MFMA", 7: "signal+stores", 8: "next item", 11: "wait free", 12: "wait loads", 6: "peak + scale", 9: "split+write", 13: "wait wrap loads", 3: "round-1 write", 10: "round-1 loads", 4: "signal", 14: "find next", 5: "loop top"}
rows = [list(map(int, l.split())) for l in open(sys.argv[1])]
items = float(sys.argv[2]) if len(sys.argv) > 2 else 111.5
a = np.array(rows, dtype=np.float64)
for w in range(16):
    m = a[a[:, 1] == w][:, 2:].mean(axis=0) / items
    parts = [f"{NAMES[t]} {m[t]:6.0f}" for t in sorted(NAMES) if m[t] > 0]
    print(f"wave {w:2d}: total {m.sum():6.0f} cyc/item | " + " | ".join(parts))
