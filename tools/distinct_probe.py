"""tools/distinct_probe.py [streams] [frames] -- a bulk batch whose streams are all in different states, launch after launch
WITHOUT reset: every launch replans every stream (no plan-cache hit); wall clock per launch next to the kernel's time."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import resampler_amd as ra
from resampler_amd import synth
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
chunk = 1024
dev = torch.device("cuda:0")
hs = [ra.ResamplerFir.new(2, ra.SampleRate.Hz44100, ra.SampleRate.Hz48000, ra.Latency.Sample64, ra.Attenuation.Db90) for _ in range(S)]
warm = np.zeros(2 * 4096, np.float32)
for i, h in enumerate(hs):
    h.resample_bulk(warm[: 2 * (64 + 37 * i)], chunk)
base = torch.from_numpy(synth.sweep(N, 2, 44100.0)).to(dev)
d_in = [base.clone() for _ in range(S)]
cap = max(h.bulk_output_bound(2 * N, chunk) for h in hs)
d_out = [torch.empty(cap, device=dev) for _ in range(S)]
b = ra.FirBatch(hs); b.bind(d_in, d_out)
st = ra.torch_stream()
hs[0].set_profiling(True)
times = []
for rep in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    b.resample_bulk_device(chunk, st)
    torch.cuda.synchronize(); times.append((time.perf_counter() - t0) * 1e3)
k_ms, _ = hs[0].mean_kernel_ms()
print("streams %d frames %d: launches (ms, every one replanned) %s ; kernel %.3f ms" % (S, N, " ".join("%.2f" % t for t in times), k_ms))
