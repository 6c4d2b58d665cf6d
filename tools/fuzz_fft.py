"""Random FFT rate pairs / channel counts / block counts through the bulk entry against the oracle (run per channel: the
reference's multi-channel scratch regions collide for many pairs), two bulk calls per case so that the
second starts from a carried overlap.  usage (GPU box): python tools/fuzz_fft.py [SECONDS] [SEED]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch

import resampler_amd as ra
from resampler_amd import synth
from oracle import pyoracle as o

R = [22050, 16000, 32000, 44100, 48000, 88200, 96000, 176400, 192000, 384000]


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    dev = torch.device("cuda:0")
    t0, rounds, worst = time.time(), 0, 0.0
    while time.time() - t0 < seconds:
        a, b = rng.choice(len(R), 2, replace=False)
        a, b = R[a], R[b]
        ch = int(rng.choice([1, 2, 2, 3, 4, 4, 6, 8, 10, 16]))
        g = ra.ResamplerFft.new(ch, ra.SampleRate(R.index(a)), ra.SampleRate(R.index(b)))
        n_in, n_out = g.chunk_size_input(), g.chunk_size_output()
        blocks = int(rng.integers(2, max(3, min(40, (1 << 21) // max(n_in, n_out)))))
        per_channel = [o.OracleFft(1, a, b) for _ in range(ch)]
        level = float(2.0 ** rng.integers(-12, 2))
        x = (synth.fast_noise(blocks * n_in, seed=int(rng.integers(1, 1 << 30))) * level).astype(np.float32)
        ref = np.zeros((blocks, n_out // ch, ch), np.float32)
        row = np.zeros(n_out // ch, np.float32)
        for k in range(blocks):
            xk = x[k * n_in:(k + 1) * n_in].reshape(-1, ch)
            for c in range(ch):
                assert per_channel[c].resample(np.ascontiguousarray(xk[:, c]), row) == 0
                ref[k, :, c] = row
        first = int(rng.integers(0, blocks))
        d_in, d_out = torch.from_numpy(x).to(dev), torch.zeros(blocks * n_out, device=dev)
        torch.cuda.synchronize()
        if first:
            g.resample_bulk_device(d_in[:first * n_in], d_out[:first * n_out], first)
        g.resample_bulk_device(d_in[first * n_in:], d_out[first * n_out:], blocks - first)
        torch.cuda.synchronize()
        y = d_out.cpu().numpy()
        e = float(np.sqrt(np.mean((y.astype(np.float64) - ref.reshape(-1)) ** 2))) / level
        worst = max(worst, e)
        assert e <= 1e-6, (a, b, ch, blocks, first, e)
        rounds += 1
    print(f"fuzz_fft: {rounds} rounds, worst rms relative to the level {worst:.3e}: OK")


if __name__ == "__main__":
    main()
