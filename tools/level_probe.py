"""tools/level_probe.py -- per-channel precision of the split kernels when the channels of a pair sit at very different
levels (the block-floating-point scale is per item and channel PAIR): channel 0 a full-scale sweep, channel 1 the same
sweep at 2^-12 / 2^-20 / 2^-30; and a full-scale burst followed by 2^-20 noise inside one 16-period item.  Prints the
RMS error per channel relative to the channel's level and relative to the pair's peak."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import resampler_amd as ra
from oracle import pyoracle as o
from resampler_amd import synth

def rms(a): return float(np.sqrt(np.mean(np.asarray(a, np.float64) ** 2)))
kind = o.CONVOLVE_AVX_FMA if o.have_avx_fma() else o.CONVOLVE_SCALAR
for ch, in_hz, out_hz in [(2, 44100, 48000), (4, 44100, 48000), (8, 96000, 44100), (2, 96000, 44100)]:
    for e in (12, 20, 30):
        n = 70000
        x = synth.sweep(n, ch, float(in_hz)).reshape(n, ch).copy()
        x[:, 1::2] *= np.float32(2.0 ** -e)
        x = x.reshape(-1)
        g = ra.ResamplerFir.new_from_hz(ch, in_hz, out_hz, ra.Latency.Sample64, ra.Attenuation.Db90)
        r = o.OracleFir(ch, in_hz, out_hz, 128, 90, kind)
        yg, _ = g.resample_bulk(x, 512 - 512 % ch)
        yr, _ = r.resample_all(x, 512 - 512 % ch)
        yg, yr = yg.reshape(-1, ch), yr.reshape(-1, ch)
        for c in (0, 1):
            err = rms(yg[:, c].astype(np.float64) - yr[:, c])
            print(f"{ch} ch {in_hz}->{out_hz} quiet 2^-{e} channel {c}: rel {err / rms(yr[:, c]):.2e}  vs pair peak {err:.2e} (2^{np.log2(max(err, 1e-300)):.1f})")
# burst then quiet inside one item
n = 70000
x = (synth.fast_noise(n * 2, seed=3) * np.float32(2.0 ** -20)).reshape(n, 2)
x[1000:1040] = synth.fast_noise(80, seed=4).reshape(40, 2)
x = x.reshape(-1).astype(np.float32)
g = ra.ResamplerFir.new_from_hz(2, 44100, 48000, ra.Latency.Sample64, ra.Attenuation.Db90)
r = o.OracleFir(2, 44100, 48000, 128, 90, kind)
yg, _ = g.resample_bulk(x, 512)
yr, _ = r.resample_all(x, 512)
d = yg.astype(np.float64) - yr
seg = slice(2 * 1400, 2 * 3400)   # outputs behind the burst, inside its item
print(f"burst + 2^-20 noise: whole rel {rms(d) / rms(yr):.2e}; behind the burst abs {rms(d[seg]):.2e} rel to the noise {rms(d[seg]) / rms(yr[seg]):.2e}")
