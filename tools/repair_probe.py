"""tools/repair_probe.py CH IN_HZ OUT_HZ [frames] -- GPU box: one bulk launch of one stream with the repair pass switched off
(RSMP_DEBUG=1 RSMP_FIR_NO_REPAIR=1) against the oracle: which channels / 1024-frame chunks the periodic kernel itself got
wrong (what the repair pass would have had to redo).  With clean audio the answer must be: none."""
import os
import sys

os.environ["RSMP_DEBUG"] = "1"
os.environ["RSMP_FIR_NO_REPAIR"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np

import resampler_amd as ra
from oracle import pyoracle as orc
from resampler_amd import synth


def main():
    ch, in_hz, out_hz = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    frames = int(sys.argv[4]) if len(sys.argv) > 4 else 100000
    x = synth.sweep(frames, ch, float(in_hz))
    gpu = ra.ResamplerFir.new_from_hz(ch, in_hz, out_hz, ra.Latency.Sample64, ra.Attenuation.Db90)
    ref = orc.OracleFir(ch, in_hz, out_hz, 128, 90)
    yg, _ = gpu.resample_bulk(x, 512 * ch)
    yr, _ = ref.resample_all(x, 512 * ch)
    print("variant", gpu.kernel_variant(), "values", yg.size, yr.size)
    n = min(yg.size, yr.size) // ch
    g = yg[:n * ch].reshape(n, ch).astype(np.float64)
    r = yr[:n * ch].reshape(n, ch).astype(np.float64)
    for c in range(ch):
        e = g[:, c] - r[:, c]
        bad = np.nonzero(~(np.abs(e) <= 1e-5))[0]
        print(f"channel {c}: rms err {np.sqrt(np.mean(np.nan_to_num(e, nan=1.0) ** 2)):.3e}, frames off by > 1e-5: {bad.size}",
              (f"first {bad[:8]} chunks {sorted(set((bad >> 10).tolist()))[:12]}" if bad.size else ""))
        if bad.size:
            i = bad[0]
            print("   gpu", g[i:i + 4, c], "ref", r[i:i + 4, c])


if __name__ == "__main__":
    main()
