"""FFT bulk throughput for channel counts other than 2 (the C2 = false instantiations of fft_ola_wave_kernel).
usage (GPU box): python tools/fft_channels_bench.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch

import resampler_amd as ra
from resampler_amd import synth


def main():
    dev = torch.device("cuda:0")
    for ch, streams, blocks in ((1, 128, 892), (2, 64, 892), (8, 16, 892), (6, 16, 892)):
        hs = [ra.ResamplerFft.new(ch, ra.SampleRate.Hz44100, ra.SampleRate.Hz48000) for _ in range(streams)]
        n_in, n_out = hs[0].chunk_size_input(), hs[0].chunk_size_output()
        x = torch.from_numpy(synth.fast_noise(blocks * n_in, seed=3)).to(dev)
        d_in = [(x * (0.5 + 0.5 * i / len(hs))).contiguous() for i in range(len(hs))]   # a buffer per stream: all bytes from HBM
        d_out = [torch.empty(blocks * n_out, device=dev) for _ in hs]
        batch = ra.FftBatch(hs)
        batch.bind(d_in, d_out, [blocks] * streams)
        s = ra.torch_stream()
        for _ in range(5):
            batch.resample_bulk_device(s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 30
        for _ in range(n):
            batch.resample_bulk_device(s)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        alg = 4.0 * (n_in + n_out) * blocks * streams
        print(f"channels {ch}: {streams} streams x {blocks} blocks: {dt * 1e3:.3f} ms per launch, "
              f"{alg / dt / 1e9:.0f} GB/s algorithmic = {alg / dt / 8e12 * 100:.1f} % of 8 TB/s")


if __name__ == "__main__":
    main()
