"""FFT bulk throughput for rate pairs other than the headline's (which run on the workgroup-per-transform
kernels).  usage (GPU box): python tools/fft_pairs_bench.py [--all | IN_HZ:OUT_HZ ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch

import resampler_amd as ra
from resampler_amd import synth

R = [22050, 16000, 32000, 44100, 48000, 88200, 96000, 176400, 192000, 384000]


def main():
    dev = torch.device("cuda:0")
    pairs = [tuple(int(v) for v in arg.split(":")) for arg in sys.argv[1:] if ":" in arg]
    if "--all" in sys.argv:
        pairs = [(a, b) for a in R for b in R if a != b]
    if not pairs:
        pairs = [(44100, 48000), (48000, 44100), (48000, 96000), (96000, 48000), (44100, 96000), (32000, 48000), (16000, 48000),
                 (22050, 44100), (88200, 96000), (192000, 48000), (44100, 384000), (48000, 192000), (48000, 16000),
                 (48000, 32000), (384000, 32000), (384000, 48000)]
    for a, b in pairs:
        streams = 64
        hs = [ra.ResamplerFft.new(2, ra.SampleRate(R.index(a)), ra.SampleRate(R.index(b))) for _ in range(streams)]
        n_in, n_out = hs[0].chunk_size_input(), hs[0].chunk_size_output()
        blocks = max(4, (1 << 21) // n_in)          # ~2^20 frames per stream
        x = torch.from_numpy(synth.fast_noise(blocks * n_in, seed=3)).to(dev)
        d_in = [(x * (0.5 + 0.5 * i / len(hs))).contiguous() for i in range(len(hs))]   # a buffer per stream: all bytes from HBM
        d_out = [torch.empty(blocks * n_out, device=dev) for _ in hs]
        batch = ra.FftBatch(hs)
        batch.bind(d_in, d_out, [blocks] * streams)
        s = ra.torch_stream()
        for _ in range(3):
            batch.resample_bulk_device(s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            batch.resample_bulk_device(s)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        alg = 4.0 * (n_in + n_out) * blocks * streams
        print(f"{a}->{b}: blocks of {n_in // 2}->{n_out // 2} frames x {blocks}: {dt * 1e3:.3f} ms, "
              f"{streams * blocks * n_in / dt / 1e9:.1f} G samples/s in, {alg / dt / 8e12 * 100:.1f} % of 8 TB/s")


if __name__ == "__main__":
    main()
