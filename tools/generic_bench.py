"""tools/generic_bench.py [IN_HZ OUT_HZ [CH [STREAMS]]] -- GPU box: bulk throughput of streams whose ratio has no short period
(ResamplerFir::new_from_hz with arbitrary rates: default 2 ch 44100 -> 47999 Hz, 64 streams x 2^20 frames, 512-value calls), the
kernel that served the launch, and one stream against the oracle.  RSMP_DEBUG=1 RSMP_FIR_GENERIC_BULK=0: the latency kernel alone."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch

import resampler_amd as ra
from oracle import pyoracle as orc
from resampler_amd import synth


def main():
    in_hz = int(sys.argv[1]) if len(sys.argv) > 1 else 44100
    out_hz = int(sys.argv[2]) if len(sys.argv) > 2 else 47999
    ch = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    streams = int(sys.argv[4]) if len(sys.argv) > 4 else max(1, 128 // ch)
    dev = torch.device("cuda:0")
    frames = 1 << 20
    hs = [ra.ResamplerFir.new_from_hz(ch, in_hz, out_hz, ra.Latency.Sample64, ra.Attenuation.Db90) for _ in range(streams)]
    x = synth.fast_noise(frames * ch, seed=3)
    d_in = [(torch.from_numpy(x).to(dev) * (0.5 + 0.5 * i / streams)).contiguous() for i in range(streams)]
    d_out = [torch.zeros(hs[0].bulk_output_bound(frames * ch, 512 * ch), device=dev) for _ in hs]
    batch = ra.FirBatch(hs)
    batch.bind(d_in, d_out)
    s = ra.torch_stream()

    def step():
        batch.reset()
        return batch.resample_bulk_device(512 * ch, s)
    cons, prod = step()
    torch.cuda.synchronize()
    # parity of the last stream against the oracle (AVX+FMA leaf where the host has it)
    kind = orc.CONVOLVE_AVX_FMA if orc.have_avx_fma() else orc.CONVOLVE_SCALAR
    ref = orc.OracleFir(ch, in_hz, out_hz, 128, 90, kind)
    yr, _ = ref.resample_all(d_in[-1].cpu().numpy(), 512 * ch)
    yg = d_out[-1].cpu().numpy()[:int(prod[-1])]
    ok = yg.size == yr.size
    err = float(np.sqrt(np.mean((yg.astype(np.float64) - yr[:yg.size]) ** 2))) / float(np.sqrt(np.mean(yr.astype(np.float64) ** 2))) if ok else float("nan")
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    alg = 4.0 * (streams * frames * ch + float(sum(prod)))
    print(f"{ch} ch {in_hz}->{out_hz}: variant {hs[0].kernel_variant()}  {dt * 1e3:.3f} ms  {streams * frames * ch / dt / 1e9:.1f} G samples/s in  "
          f"{alg / dt / 8e12 * 100:.2f} % of 8 TB/s   counts {'equal' if ok else 'DIFFER'}  rel rms err {err:.2e}")


if __name__ == "__main__":
    main()
