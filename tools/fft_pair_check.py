"""One FFT rate pair against the oracle (run per channel: the reference's multi-channel scratch regions collide for
many pairs) through the bulk entry.  usage (GPU box):
python tools/fft_pair_check.py IN_HZ OUT_HZ CHANNELS BLOCKS [...more quadruples]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch

import resampler_amd as ra
from resampler_amd import synth
from oracle import pyoracle as o

R = [22050, 16000, 32000, 44100, 48000, 88200, 96000, 176400, 192000, 384000]


def main():
    dev = torch.device("cuda:0")
    args = [int(v) for v in sys.argv[1:]]
    for a, b, ch, blocks in zip(args[0::4], args[1::4], args[2::4], args[3::4]):
        g = ra.ResamplerFft.new(ch, ra.SampleRate(R.index(a)), ra.SampleRate(R.index(b)))
        per_channel = [o.OracleFft(1, a, b) for _ in range(ch)]
        n_in, n_out = g.chunk_size_input(), g.chunk_size_output()
        x = synth.fast_noise(blocks * n_in, seed=5)
        d_out = torch.zeros(blocks * n_out, device=dev)
        d_in = torch.from_numpy(x).to(dev)
        torch.cuda.synchronize()   # (the launch runs on the handle's own stream: order it after the fills)
        g.resample_bulk_device(d_in, d_out, blocks)
        torch.cuda.synchronize()
        ref3 = np.zeros((blocks, n_out // ch, ch), np.float32)
        row = np.zeros(n_out // ch, np.float32)
        for k in range(blocks):
            xk = x[k * n_in:(k + 1) * n_in].reshape(-1, ch)
            for c in range(ch):
                assert per_channel[c].resample(np.ascontiguousarray(xk[:, c]), row) == 0
                ref3[k, :, c] = row
        ref = ref3.reshape(blocks, n_out)
        y = d_out.cpu().numpy().reshape(blocks, n_out)
        per_block = np.sqrt(np.mean((y.astype(np.float64) - ref) ** 2, axis=1))
        bad = [int(k) for k in np.nonzero(per_block > 1e-6)[0]]
        print(f"{a}->{b} ch={ch} blocks={blocks}: worst block rms {per_block.max():.3e}, bad blocks {bad[:12]}")
        if bad:
            k = bad[0]
            d = np.abs(y[k] - ref[k]).reshape(-1, ch)
            idx = np.nonzero(d.max(axis=1) > 1e-5)[0]
            f0 = int(idx[0])
            print("   got", y[k].reshape(-1, ch)[f0:f0 + 4, 0], "want", ref[k].reshape(-1, ch)[f0:f0 + 4, 0], "got-want",
                  (y[k] - ref[k]).reshape(-1, ch)[f0:f0 + 4, 0])
            print(f"   block {k}: {idx.size} bad frames of {n_out // ch}, first {idx[:8]}, last {idx[-4:]}, per channel {(d > 1e-5).sum(axis=0)}")


if __name__ == "__main__":
    main()
