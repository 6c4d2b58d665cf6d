#!/usr/bin/env python3
"""Phase clocks of fir_lockstep_kernel (diagnostic instantiation, RSMP_LS_TRACE): one traced step of config 4.
Columns per workgroup: wave 0 = planned, columns built, past barrier, units done, wraps done, past barrier 2, end;
wave 1 = (unused), staged, past barrier, units done, ...  (shader-clock cycles since kernel entry)."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RSMP_DEBUG", "1")   # (the library reads its diagnostic switches only then)
path = os.environ.get("RSMP_LS_TRACE") or os.path.join(ROOT, "gpurun_out", "ls_trace.txt")
if os.path.exists(path):
    os.remove(path)
os.environ["RSMP_LS_TRACE"] = path
import torch
import resampler_amd as ra
from resampler_amd import sharding, synth
dev = torch.device("cuda:0")
specs = sharding.mixed_rate_batch(1024, 2, 512)
hs = [ra.ResamplerFir.new_from_hz(2, s.in_hz, s.out_hz, ra.Latency.Sample64, ra.Attenuation.Db90) for s in specs]
x = torch.from_numpy(synth.fast_noise(8 * 512 * 2, seed=1)).to(dev)
d_in = [x.clone() for _ in specs]
caps = [h.buffer_size_output() for h in hs]
d_out = [torch.empty(c, device=dev) for c in caps]
ls = ra.FirLockstep(hs, 512)
ls.bind_caps(d_in, d_out, caps)
for k in range(4):
    ls.step(512, (k % 8) * 512)
ls.counts()
rows = np.loadtxt(path)
rows = rows[-ls.workgroups():]          # the last traced step
names = ["planned", "cols", "barrier1", "units", "wraps", "barrier2", "end"]
print("workgroups", rows.shape[0])
for w in range(2):
    blk = rows[:, 1 + 7 * w: 8 + 7 * w]
    print(f"wave {w}: " + "  ".join(f"{n} p50 {np.percentile(blk[:, i], 50):.0f} max {blk[:, i].max():.0f}" for i, n in enumerate(names)))
ua = rows[:, 15:20]
n = np.maximum(ua[:, 4], 1)
print("wave 1 per unit: units p50 %.1f | setup %.0f | tile wait %.0f | mfma %.0f | epilogue %.0f (cycles, p50 over workgroups)" % (
    np.percentile(ua[:, 4], 50), np.percentile(ua[:, 0] / n, 50), np.percentile(ua[:, 1] / n, 50), np.percentile(ua[:, 2] / n, 50), np.percentile(ua[:, 3] / n, 50)))
# by geometry: group rows by their 'end'
order = np.argsort(rows[:, 7])
print("slowest workgroups:", rows[order[-5:], 0].astype(int), rows[order[-5:], 7])
# end time by workgroup index (the batch's internal order groups the streams by geometry: buckets of 20 workgroups)
end = rows[:, 7]
print("end by workgroup bucket (p50 / max, k cycles):", " ".join(f"{b * 20}:{np.percentile(end[b * 20:(b + 1) * 20], 50) / 1e3:.0f}/{end[b * 20:(b + 1) * 20].max() / 1e3:.0f}" for b in range((len(end) + 19) // 20)))
stage = rows[:, 3]
print("barrier1 by bucket (p50):", " ".join(f"{b * 20}:{np.percentile(stage[b * 20:(b + 1) * 20], 50) / 1e3:.0f}" for b in range((len(end) + 19) // 20)))
