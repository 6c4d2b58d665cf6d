#!/usr/bin/env python3
"""tools/split_geo_check.py [ch in_hz out_hz]... -- bulk parity of single rate pairs against the oracle (GPU box), one line
each: kernel variant, RMS error, counts equal.  For bringing up new split-kernel geometries under a short `timeout`."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import resampler_amd as ra
from oracle import pyoracle as o
from resampler_amd import synth

args = [int(a) for a in sys.argv[1:]] or [2, 44100, 96000, 2, 48000, 96000, 2, 96000, 44100, 2, 96000, 48000]
for k in range(0, len(args), 3):
    ch, in_hz, out_hz = args[k:k + 3]
    for n in (3000, 60000, 400000):
        g = ra.ResamplerFir.new_from_hz(ch, in_hz, out_hz, ra.Latency.Sample64, ra.Attenuation.Db90)
        g.set_kernel(ra.FirKernel.Periodic)
        r = o.OracleFir(ch, in_hz, out_hz, 128, 90, o.CONVOLVE_AVX_FMA if o.have_avx_fma() else o.CONVOLVE_SCALAR)
        x = synth.sweep(n, ch, float(in_hz))
        chunk = 512 - 512 % ch
        yg, consumed, calls_g = g.resample_bulk(x, chunk, want_calls=True)
        yr, calls_r = r.resample_all(x, chunk)
        e = float(np.sqrt(np.mean((yg.astype(np.float64) - yr) ** 2))) if yg.size == yr.size and yr.size else -1.0
        bad = int(np.sum(np.abs(yg.astype(np.float64) - yr) > 1e-5)) if yg.size == yr.size else -1
        print(f"{ch} ch {in_hz}->{out_hz} n={n}: variant {g.kernel_variant()} sizes {yg.size}/{yr.size} counts_equal {np.array_equal(calls_g, calls_r)} rms {e:.3e} bad {bad}", flush=True)
