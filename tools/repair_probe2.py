"""tools/repair_probe2.py CH IN_HZ OUT_HZ [streams] -- GPU box: tools/channels_bench.py's batch (noise, a gain per stream) in one
bulk launch with the repair pass switched off, every stream against the oracle: where the periodic kernel's own output is
wrong (frames off by more than 1e-5), by 1024-frame chunk."""
import os
import sys

os.environ["RSMP_DEBUG"] = "1"
os.environ["RSMP_FIR_NO_REPAIR"] = "1"
os.environ["RSMP_FIR_COUNT_MARKS"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch

import resampler_amd as ra
from oracle import pyoracle as orc
from resampler_amd import synth


def main():
    ch, in_hz, out_hz = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    streams = int(sys.argv[4]) if len(sys.argv) > 4 else max(1, 128 // ch)
    dev = torch.device("cuda:0")
    frames = 1 << 20
    hs = [ra.ResamplerFir.new_from_hz(ch, in_hz, out_hz, ra.Latency.Sample64, ra.Attenuation.Db90) for _ in range(streams)]
    x = synth.fast_noise(frames * ch, seed=2)
    gains = [np.float32(0.5 + 0.5 * i / len(hs)) for i in range(len(hs))]
    d_in = [(torch.from_numpy(x).to(dev) * float(g)).contiguous() for g in gains]
    d_out = [torch.zeros(hs[0].bulk_output_bound(frames * ch, 512 * ch), device=dev) for _ in hs]
    batch = ra.FirBatch(hs)
    batch.bind(d_in, d_out)
    cons, prod = batch.resample_bulk_device(512 * ch, ra.torch_stream())
    torch.cuda.synchronize()
    for i in (0, 1, len(hs) - 1):
        ref = orc.OracleFir(ch, in_hz, out_hz, 128, 90)
        yr, _ = ref.resample_all(d_in[i].cpu().numpy(), 512 * ch)
        yg = d_out[i].cpu().numpy()[:int(prod[i])]
        n = min(yg.size, yr.size) // ch
        e = (yg[:n * ch].astype(np.float64) - yr[:n * ch]).reshape(n, ch)
        for c in range(ch):
            bad = np.nonzero(~(np.abs(e[:, c]) <= 1e-5))[0]
            print(f"stream {i} channel {c}: rms {np.sqrt(np.mean(np.nan_to_num(e[:, c], nan=1.0) ** 2)):.3e}  bad frames {bad.size}",
                  f"chunks {sorted(set((bad >> 10).tolist()))[:16]}" if bad.size else "")


if __name__ == "__main__":
    main()
