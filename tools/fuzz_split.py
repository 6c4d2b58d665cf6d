"""Random streams of the split kernel's rate pairs (44.1 <-> 48, 44.1 <-> 96, 48 <-> 96, 88.2 -> 44.1, 192 -> 48 kHz ...; 1 .. 16
channels, all tap counts and attenuations, 1-3 bulk launches of random length and chunk size each; lengths up to a few
hundred thousand frames so that workgroups get several items; a random level between 2^-30 and 2^6 per stream, a jump of
the level inside some) against the AVX+FMA oracle: counts identical, RMS within 1e-6 RELATIVE to the signal.
usage (GPU box): python tools/fuzz_split.py [rounds]"""
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
    pairs = [(44100, 48000), (48000, 44100), (44100, 96000), (96000, 44100), (48000, 96000), (96000, 48000), (88200, 44100),
             (192000, 48000), (22050, 48000), (88200, 96000), (44100, 88200)]
    a, b = pairs[int(rng.integers(len(pairs)))]
    lat = list(ra.Latency)[int(rng.integers(len(list(ra.Latency))))]
    att = [ra.Attenuation.Db60, ra.Attenuation.Db90, ra.Attenuation.Db120][int(rng.integers(3))]
    ch = int(rng.integers(1, 17)) if rng.integers(3) else 2
    g = ra.ResamplerFir.new_from_hz(ch, a, b, lat, att)
    g.set_kernel(ra.FirKernel.Periodic)
    r = o.OracleFir(ch, a, b, lat.taps(), {ra.Attenuation.Db60: 60, ra.Attenuation.Db90: 90, ra.Attenuation.Db120: 120}[att],
                    o.CONVOLVE_AVX_FMA if o.have_avx_fma() else o.CONVOLVE_SCALAR)
    level = np.float32(2.0 ** float(rng.integers(-30, 7)))
    for step in range(int(rng.integers(1, 4))):
        n = int(rng.integers(1, 60000)) if rng.integers(4) else int(rng.integers(1, 400))
        if rng.integers(8) == 0:
            n = int(rng.integers(100000, 400000))
        chunk = int(rng.integers(1, 300)) * ch
        x = synth.fast_noise(ch * n, seed=int(rng.integers(1 << 30))) * level
        if rng.integers(4) == 0 and n > 10:   # the level jumps inside the launch (one channel only, sometimes)
            k = int(rng.integers(1, n)) * ch
            x[k + (int(rng.integers(ch)) if rng.integers(2) else 0)::(ch if rng.integers(2) else 1)] *= np.float32(2.0 ** float(rng.integers(-12, 5)))
        x = x.astype(np.float32)
        yg, _ = g.resample_bulk(x, chunk)
        yr, _ = r.resample_all(x, chunk)
        assert yg.size == yr.size, (it, step, ch, n, chunk, yg.size, yr.size)
        if yg.size:
            e = float(np.sqrt(np.mean((yg.astype(np.float64) - yr) ** 2))) / max(float(np.sqrt(np.mean(yr.astype(np.float64) ** 2))), 1e-300)
            worst = max(worst, e)
            assert e <= 1e-6, (it, step, ch, a, b, lat, att, n, chunk, float(level), e)
        n4 += g.kernel_variant() in (4, 5)
print("fuzz ok: worst relative rms %.3e, split-kernel launches %d of %d rounds" % (worst, n4, rounds))
