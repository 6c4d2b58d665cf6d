"""Bulk FIR throughput for channel counts / rate pairs other than the headline's (which periodic kernel and
geometry they get).  usage (GPU box): [RSMP_FIR_CG=1|2] python tools/channels_bench.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch

import resampler_amd as ra
from resampler_amd import synth


def main():
    dev = torch.device("cuda:0")
    frames = 1 << 20
    for ch, in_hz, out_hz in ((2, 44100, 96000), (2, 96000, 44100), (2, 48000, 96000), (2, 96000, 48000), (2, 48000, 44100),
                              (1, 48000, 44100), (3, 44100, 48000), (4, 96000, 44100), (4, 44100, 48000), (6, 44100, 48000),
                              (6, 96000, 44100), (8, 96000, 44100), (8, 44100, 48000), (8, 48000, 96000)):
        streams = max(1, 128 // ch)
        hs = [ra.ResamplerFir.new_from_hz(ch, in_hz, out_hz, ra.Latency.Sample64, ra.Attenuation.Db90) for _ in range(streams)]
        x = torch.from_numpy(synth.fast_noise(frames * ch, seed=2)).to(dev)
        d_in = [(x * (0.5 + 0.5 * i / len(hs))).contiguous() for i in range(len(hs))]   # a buffer per stream: all bytes from HBM
        d_out = [torch.empty(hs[0].bulk_output_bound(frames * ch, 512 * ch), device=dev) for _ in hs]
        batch = ra.FirBatch(hs)
        batch.bind(d_in, d_out)
        s = ra.torch_stream()

        def step():
            batch.reset()
            return batch.resample_bulk_device(512 * ch, s)
        for _ in range(3):
            cons, prod = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        alg = 4.0 * (streams * frames * ch + float(sum(prod)))
        print(f"{ch} ch {in_hz}->{out_hz}: variant {hs[0].kernel_variant()}  {dt * 1e3:.3f} ms  {streams * frames * ch / dt / 1e9:.1f} G samples/s in  "
              f"{alg / dt / 8e12 * 100:.1f} % of 8 TB/s")


if __name__ == "__main__":
    main()
