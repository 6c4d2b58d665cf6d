import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, torch, numpy as np
import resampler_amd as ra
from resampler_amd import synth
S, N = 64, 1 << 20
dev = torch.device("cuda:0")
hs = [ra.ResamplerFir.new(2, ra.SampleRate.Hz44100, ra.SampleRate.Hz48000, ra.Latency.Sample64, ra.Attenuation.Db90) for _ in range(S)]
base = torch.from_numpy(synth.sweep(N, 2, 44100.0)).to(dev)
d_in = [base.clone() for _ in range(S)]
cap = hs[0].bulk_output_bound(2 * N, 512)
d_out = [torch.empty(cap, device=dev) for _ in range(S)]
b = ra.FirBatch(hs); b.bind(d_in, d_out)
st = ra.torch_stream()
for _ in range(50):
    b.reset(); b.resample_bulk_device(512, st)
torch.cuda.synchronize()
tr = tb = 0.0
K = 300
for _ in range(K):
    t0 = time.perf_counter(); b.reset(); t1 = time.perf_counter(); b.resample_bulk_device(512, st); t2 = time.perf_counter()
    tr += t1 - t0; tb += t2 - t1
    if _ % 50 == 49: torch.cuda.synchronize()
torch.cuda.synchronize()
print("back to back: reset %.1f us  bulk %.1f us per step" % (tr / K * 1e6, tb / K * 1e6))
tb = 0.0
for _ in range(100):
    torch.cuda.synchronize()
    b.reset(); t1 = time.perf_counter(); b.resample_bulk_device(512, st); t2 = time.perf_counter()
    tb += t2 - t1
print("GPU idle at call time: bulk %.1f us per step (pure host cost)" % (tb / 100 * 1e6))
