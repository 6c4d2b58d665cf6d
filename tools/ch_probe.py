"""tools/ch_probe.py CH IN_HZ OUT_HZ [steps] -- one channel count / rate pair of tools/channels_bench.py alone (for a
`rocprofv3 --kernel-trace --stats` run: which kernels a bulk launch of that shape is made of and what each takes)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch

import resampler_amd as ra
from resampler_amd import synth


def main():
    ch, in_hz, out_hz = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    dev = torch.device("cuda:0")
    frames = 1 << 20
    streams = max(1, 128 // ch)
    hs = [ra.ResamplerFir.new_from_hz(ch, in_hz, out_hz, ra.Latency.Sample64, ra.Attenuation.Db90) for _ in range(streams)]
    x = torch.from_numpy(synth.fast_noise(frames * ch, seed=2)).to(dev)
    d_in = [(x * (0.5 + 0.5 * i / len(hs))).contiguous() for i in range(len(hs))]
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
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    alg = 4.0 * (streams * frames * ch + float(sum(prod)))
    print(f"{ch} ch {in_hz}->{out_hz}: variant {hs[0].kernel_variant()}  {dt * 1e3:.3f} ms  {alg / dt / 8e12 * 100:.1f} % of 8 TB/s")


if __name__ == "__main__":
    main()
