"""Random 44.1 <-> 48 kHz streams (1 .. 16 channels, all tap counts and attenuations, 1-3 bulk launches of random length
and chunk size each; lengths up to a few hundred thousand frames so that workgroups get several items) against the
oracle: counts identical, RMS within 1e-6.  usage (GPU box): python tools/fuzz_split.py [rounds]"""
import sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
import resampler_amd as ra
from oracle import pyoracle as o
from resampler_amd import synth
rng = np.random.default_rng(2026)
worst = 0.0; n4 = 0
lats = [ra.Latency.Sample64]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 120
for it in range(rounds):
    a, b = (44100, 48000) if rng.integers(2) else (48000, 44100)
    lat = list(ra.Latency)[int(rng.integers(len(list(ra.Latency))))]
    att = [ra.Attenuation.Db60, ra.Attenuation.Db90, ra.Attenuation.Db120][int(rng.integers(3))]
    ch = int(rng.integers(1, 17)) if rng.integers(3) else 2
    g = ra.ResamplerFir.new_from_hz(ch, a, b, lat, att)
    g.set_kernel(ra.FirKernel.Periodic)
    r = o.OracleFir(ch, a, b, lat.taps(), {ra.Attenuation.Db60: 60, ra.Attenuation.Db90: 90, ra.Attenuation.Db120: 120}[att])
    for step in range(int(rng.integers(1, 4))):
        n = int(rng.integers(1, 60000)) if rng.integers(4) else int(rng.integers(1, 400))
        if rng.integers(8) == 0:
            n = int(rng.integers(100000, 400000))
        chunk = int(rng.integers(1, 300)) * ch
        x = synth.fast_noise(ch * n, seed=int(rng.integers(1 << 30)))
        yg, _ = g.resample_bulk(x, chunk)
        yr, _ = r.resample_all(x, chunk)
        assert yg.size == yr.size, (it, step, ch, n, chunk, yg.size, yr.size)
        if yg.size:
            e = float(np.sqrt(np.mean((yg.astype(np.float64) - yr) ** 2)))
            worst = max(worst, e)
            assert e <= 1e-6, (it, step, ch, a, b, lat, att, n, chunk, e)
        n4 += g.kernel_variant() in (4, 5)
print("fuzz ok: worst rms %.3e, split-kernel launches %d" % (worst, n4))
