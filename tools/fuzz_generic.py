"""Random ResamplerFir streams of ARBITRARY rates (ratios without a short period: the launches fir_generic_bulk.hip takes when they are long,
fir_generic_kernel when short) through the bulk entry, two or three launches per stream (the later ones on the history the earlier
left), against the oracle: counts of every call identical, 1e-6 RMS relative to the signal.
usage (GPU box): python tools/fuzz_generic.py [seconds] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np

import resampler_amd as ra
from oracle import pyoracle as orc
from resampler_amd import synth


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    kind = orc.CONVOLVE_AVX_FMA if orc.have_avx_fma() else orc.CONVOLVE_SCALAR
    lat = {16: ra.Latency.Sample8, 32: ra.Latency.Sample16, 64: ra.Latency.Sample32, 128: ra.Latency.Sample64}
    t0 = time.time()
    rounds, worst, bulk_launches = 0, 0.0, 0
    while time.time() - t0 < seconds:
        ch = int(rng.choice([1, 1, 2, 2, 2, 3, 4, 5, 6, 8]))
        taps = int(rng.choice([16, 32, 64, 128, 128, 128]))
        a = int(rng.integers(7000, 200000)) | 1
        b = int(rng.integers(7000, 200000)) | 1
        if a == b or max(a, b) / min(a, b) > 6.0:
            continue
        att = int(rng.choice([60, 90, 120]))
        gpu = ra.ResamplerFir.new_from_hz(ch, a, b, lat[taps], {60: ra.Attenuation.Db60, 90: ra.Attenuation.Db90, 120: ra.Attenuation.Db120}[att])
        ref = orc.OracleFir(ch, a, b, taps, att, kind)
        level = float(2.0 ** rng.integers(-12, 3))
        for _ in range(int(rng.integers(2, 4))):
            frames = int(rng.integers(300, 70000))
            x = (synth.fast_noise(frames * ch, seed=int(rng.integers(1 << 30))) * np.float32(level)).astype(np.float32)
            chunk = int(rng.choice([512, 1024, 300, 4096])) // ch * ch or ch
            yg, consumed, calls_g = gpu.resample_bulk(x, chunk, want_calls=True)
            yr, calls_r = ref.resample_all(x, chunk)
            assert yg.size == yr.size and np.array_equal(calls_g, calls_r), (ch, a, b, taps, frames, chunk)
            if yr.size:
                e = float(np.sqrt(np.mean((yg.astype(np.float64) - yr) ** 2))) / max(float(np.sqrt(np.mean(yr.astype(np.float64) ** 2))), 1e-30)
                assert e <= 1e-6, (ch, a, b, taps, frames, chunk, e)
                worst = max(worst, e)
            if gpu.kernel_variant() == 0 and yr.size // ch >= 8192:
                bulk_launches += 1
        rounds += 1
    print(f"fuzz_generic: {rounds} streams, {bulk_launches} launches long enough for the tiled kernel, worst relative rms {worst:.3e}, seed {seed}: OK")


if __name__ == "__main__":
    main()
