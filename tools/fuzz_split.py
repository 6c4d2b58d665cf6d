import sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
import resampler_amd as ra
from oracle import pyoracle as o
from resampler_amd import synth
rng = np.random.default_rng(2026)
worst = 0.0; n4 = 0
lats = [ra.Latency.Sample64]
for it in range(120):
    a, b = (44100, 48000) if rng.integers(2) else (48000, 44100)
    lat = list(ra.Latency)[int(rng.integers(len(list(ra.Latency))))]
    att = [ra.Attenuation.Db60, ra.Attenuation.Db90, ra.Attenuation.Db120][int(rng.integers(3))]
    g = ra.ResamplerFir.new_from_hz(2, a, b, lat, att)
    g.set_kernel(ra.FirKernel.Periodic)
    r = o.OracleFir(2, a, b, lat.taps(), {ra.Attenuation.Db60: 60, ra.Attenuation.Db90: 90, ra.Attenuation.Db120: 120}[att])
    for step in range(int(rng.integers(1, 4))):
        n = int(rng.integers(1, 60000)) if rng.integers(4) else int(rng.integers(1, 400))
        chunk = int(rng.integers(1, 300)) * 2
        x = synth.fast_noise(2 * n, seed=int(rng.integers(1 << 30)))
        yg, _ = g.resample_bulk(x, chunk)
        yr, _ = r.resample_all(x, chunk)
        assert yg.size == yr.size, (it, step, n, chunk, yg.size, yr.size)
        if yg.size:
            e = float(np.sqrt(np.mean((yg.astype(np.float64) - yr) ** 2)))
            worst = max(worst, e)
            assert e <= 1e-6, (it, step, a, b, lat, att, n, chunk, e)
        n4 += g.kernel_variant() == 4
print("fuzz ok: worst rms %.3e, split-kernel launches %d" % (worst, n4))
