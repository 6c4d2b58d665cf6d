#!/usr/bin/env python3
"""Diagnostic for the split-bf16 FIR kernel: where does a launch differ from the oracle?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import resampler_amd as ra
from oracle import pyoracle as o
from resampler_amd import synth

in_hz, out_hz = int(sys.argv[1]), int(sys.argv[2])
g = ra.ResamplerFir.new_from_hz(2, in_hz, out_hz, ra.Latency.Sample64, ra.Attenuation.Db90)
g.set_kernel(ra.FirKernel.Periodic)
r = o.OracleFir(2, in_hz, out_hz, ra.Latency.Sample64.taps(), 90)
for step, n in enumerate([60000, 3000, 6000, 20011]):
    x = synth.fast_noise(2 * n, seed=5 + step)
    yg, _ = g.resample_bulk(x, 512)
    yr, _ = r.resample_all(x, 512)
    print("step", step, "frames", n, "variant", g.kernel_variant(), "sizes", yg.size, yr.size)
    e = yg.astype(np.float64) - yr.astype(np.float64)
    bad = np.flatnonzero(~np.isfinite(e) | (np.abs(e) > 1e-5))
    print("  rms", np.sqrt(np.nanmean(e ** 2)), "bad", bad.size, "nan", int(np.isnan(yg).sum()))
    if bad.size:
        fr = bad // 2
        print("  bad frames: first", fr[:12], "last", fr[-5:])
        b = out_hz // np.gcd(in_hz, out_hz)
        print("  classes of bad (frame idx mod b is launch-relative only)", np.unique(fr % b)[:40])
        print("  yg", yg[bad[:6]], "yr", yr[bad[:6]])
