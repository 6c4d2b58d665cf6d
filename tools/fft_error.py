"""RMS distance of the GPU FFT path from the oracle (the 1e-6 gate of tests/test_fft_gpu.py) on BASELINE
config 3's rate pair; also against a float64 evaluation of the same overlap-add pipeline, to see whether a
change of arithmetic moves the result towards or away from the exact answer.
usage (GPU box): python tools/fft_error.py [blocks]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch

import resampler_amd as ra
from oracle import pyoracle as o
from resampler_amd import synth


def rms(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


def main():
    blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    for in_hz, out_hz, a, b in ((44100, 48000, ra.SampleRate.Hz44100, ra.SampleRate.Hz48000),
                                (48000, 44100, ra.SampleRate.Hz48000, ra.SampleRate.Hz44100)):
        g = ra.ResamplerFft.new(2, a, b)
        n_in, n_out = g.chunk_size_input(), g.chunk_size_output()
        x = synth.sweep(blocks * n_in // 2, 2, float(in_hz))
        dev = torch.device("cuda:0")
        d_in, d_out = [torch.from_numpy(x).to(dev)], [torch.zeros(blocks * n_out, device=dev)]
        batch = ra.FftBatch([g])
        batch.bind(d_in, d_out, [blocks])
        batch.resample_bulk_device(ra.torch_stream())
        torch.cuda.synchronize()
        y = d_out[0].cpu().numpy()
        r = o.OracleFft(2, in_hz, out_hz)
        ref = np.zeros((blocks, n_out), np.float32)
        for k in range(blocks):
            assert r.resample(x[k * n_in:(k + 1) * n_in], ref[k]) == 0
        ref = ref.reshape(-1)
        print(f"{in_hz}->{out_hz}: rms(gpu, oracle) = {rms(y, ref):.3e}  max = {np.abs(y - ref).max():.3e}  "
              f"signal rms = {float(np.sqrt(np.mean(ref.astype(np.float64) ** 2))):.3f}  variant = {g.kernel_variant() if hasattr(g, 'kernel_variant') else '-'}")


if __name__ == "__main__":
    main()
