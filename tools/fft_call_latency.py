import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import torch, resampler_amd as ra
from resampler_amd import synth
g = ra.ResamplerFft.new(2, ra.SampleRate.Hz44100, ra.SampleRate.Hz48000)
n_in, n_out = g.chunk_size_input(), g.chunk_size_output()
x = synth.sweep(n_in // 2, 2, 44100.0)
o = np.zeros(n_out, np.float32)
for _ in range(50): g.resample(x, o)
t = []
for _ in range(400):
    t0 = time.perf_counter(); g.resample(x, o); t.append(time.perf_counter() - t0)
t = np.array(t) * 1e6
print("per-call host-slice resample (1 block, 2 ch): p50 %.1f us  p90 %.1f us" % (np.percentile(t, 50), np.percentile(t, 90)))
dev = torch.device("cuda:0")
dx = torch.from_numpy(x).to(dev); do = torch.zeros(n_out, device=dev)
for _ in range(50): g.resample_bulk_device(dx, do, 1, ra.torch_stream())
torch.cuda.synchronize()
t = []
for _ in range(400):
    t0 = time.perf_counter(); g.resample_bulk_device(dx, do, 1, ra.torch_stream()); torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
t = np.array(t) * 1e6
print("device call + sync (1 block): p50 %.1f us  p90 %.1f us" % (np.percentile(t, 50), np.percentile(t, 90)))
