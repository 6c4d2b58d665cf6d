"""tools/pcm_probe.py -- is the two-round split kernel bit-reproducible run to run, and how far is the PCM build from it?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import resampler_amd as ra
from resampler_amd import synth
dev = torch.device("cuda:0")
in_hz, out_hz, frames, bits = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
x = synth.sweep(frames, 2, float(in_hz)) * 0.9
s = np.clip(np.round(x.astype(np.float64) * (1 << (bits - 1))), -(1 << (bits - 1)), (1 << (bits - 1)) - 1).astype(np.int64)
raw = s.astype("<i2").tobytes() if bits == 16 else s.astype("<i4").tobytes()
d_pcm = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
d_f32 = torch.empty(2 * frames, device=dev)
ra.pcm_to_stereo_f32_device(d_pcm, bits, 2, d_f32)
outs = []
for mode in ("f32", "f32", "pcm"):
    h = ra.ResamplerFir.new_from_hz(2, in_hz, out_hz, ra.Latency.Sample64, ra.Attenuation.Db90)
    cap = h.bulk_output_bound(2 * frames, 512)
    o = torch.zeros(cap, device=dev)
    b = ra.FirBatch([h])
    if mode == "f32":
        b.bind([d_f32], [o]); c, p = b.resample_bulk_device(512)
    else:
        c, p = b.resample_bulk_pcm_device([d_pcm], bits, [o], 512)
    torch.cuda.synchronize()
    outs.append(o[:int(p[0])].clone())
a, b_, c = outs
print("f32 vs f32 equal:", torch.equal(a, b_), "max diff", float((a - b_).abs().max()))
d = (a - c).abs()
print("f32 vs pcm equal:", torch.equal(a, c), "max diff", float(d.max()), "n diff", int((d > 0).sum()), "of", d.numel(), "first idx", int(torch.nonzero(d > 0)[0]) if (d > 0).any() else -1)
