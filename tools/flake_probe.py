#!/usr/bin/env python3
"""Hunts an intermittent mismatch of the 4-channel vector kernel's per-call path (diagnostic)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import resampler_amd as ra
from oracle import pyoracle as o
from resampler_amd import synth

bad_total = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    g2 = ra.ResamplerFir.new_from_hz(2, 44100, 48000, ra.Latency.Sample64, ra.Attenuation.Db90)
    g2.set_kernel(ra.FirKernel.Periodic)
    g2.resample_bulk(synth.sweep(60000 + 7 * it, 2, 44100.0), 512)
    ch = 4
    g = ra.ResamplerFir.new_from_hz(ch, 22050, 48000, ra.Latency.Sample64, ra.Attenuation.Db60)
    g.set_kernel(ra.FirKernel.PeriodicVector)
    r = o.OracleFir(ch, 22050, 48000, ra.Latency.Sample64.taps(), 60)
    x = synth.sweep(60000, ch, 22050.0)
    yg, _ = g.resample_bulk(x, 512 - 512 % ch)
    yr, _ = r.resample_all(x, 512 - 512 % ch)
    e0 = float(np.sqrt(np.mean((yg.astype(np.float64) - yr) ** 2)))
    x2 = synth.fast_noise(ch * 3000, seed=5)
    og = np.zeros(g.buffer_size_output(), np.float32)
    orr = np.zeros(r.buffer_size_output(), np.float32)
    off = 0; call = 0
    while off < x2.size:
        sl = x2[off:off + 700 * ch]
        cg, pg = g.resample(sl, og)
        rc, cr, pr = r.resample(sl, orr)
        err = np.abs(og[:pg].astype(np.float64) - orr[:pr])
        if (cg, pg) != (cr, pr) or (err.size and err.max() > 1e-4):
            badi = np.flatnonzero(err > 1e-4)
            print(f"iter {it} call {call}: counts {(cg,pg)} vs {(cr,pr)} bulk_rms {e0:.2e} bad {badi.size}/{pg} first {badi[:6]//ch} last {badi[-3:]//ch} variant {g.kernel_variant()} got {og[badi[:3]]} want {orr[badi[:3]]}")
            bad_total += 1
            break
        off += cg; call += 1
        if cg == 0 and pg == 0: break
print("bad iterations:", bad_total)
