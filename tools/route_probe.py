import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import resampler_amd as ra
from resampler_amd import synth
S, N, chunk = 64, 1 << 20, 512
dev = torch.device("cuda:0")
hs = [ra.ResamplerFir.new(2, ra.SampleRate.Hz44100, ra.SampleRate.Hz48000, ra.Latency.Sample64, ra.Attenuation.Db90) for _ in range(S)]
warm = np.zeros(2 * 4096, np.float32)
for i, h in enumerate(hs):
    h.resample_bulk(warm[: 2 * (64 + 37 * i)], 512)
base = torch.from_numpy(synth.sweep(N, 2, 44100.0)).to(dev)
d_in = [base.clone() for _ in range(S)]
d_out = [torch.empty(h.bulk_output_bound(2 * N, chunk), device=dev) for h in hs]
b = ra.FirBatch(hs); b.bind(d_in, d_out)
fresh = os.environ.get("PROBE_FRESH")   # other buffers for every launch (two sets, in turn): the library re-binds its batch without losing the run planned ahead
if fresh:
    sets = [(d_in, d_out), ([t.clone() for t in d_in], [torch.empty_like(t) for t in d_out])]
st = torch.cuda.Stream()
ts, tc = [], []
for rep in range(10):
    if fresh:
        b.bind(*sets[rep % 2])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    b.resample_bulk_device(chunk, st.cuda_stream)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3); tc.append((t1 - t0) * 1e3)
print("routed launches (ms):", " ".join("%.2f" % t for t in ts)); print("  the call itself:", " ".join("%.2f" % t for t in tc))
