import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import resampler_amd as ra
from resampler_amd import sharding, synth
n = int(sys.argv[1]); k = int(sys.argv[2]); steps = int(sys.argv[3]); frames = 512
dev = torch.device("cuda:0")
specs = sharding.mixed_rate_batch(n, 2, frames)
hs = [ra.ResamplerFir.new_from_hz(2, s.in_hz, s.out_hz, ra.Latency.Sample64, ra.Attenuation.Db90) for s in specs]
caps = [h.buffer_size_output() for h in hs]
d_in = [torch.from_numpy(synth.hash_noise(steps * frames * 2, seed=i)).to(dev) for i in range(n)]
room = [steps * 2 * (frames * s.out_hz // s.in_hz + 2) + caps[i] for i, s in enumerate(specs)]
d_out = [torch.zeros(room[i], device=dev) for i in range(n)]
ls = ra.FirLockstep(hs, frames)
ls.bind_caps(d_in, d_out, caps)
for s0 in range(0, steps, k):
    ls.run(k, frames, s0 * frames, append=True)
    cons, prod = ls.run_counts()
    print("run at", s0, "ok; produced", int(prod.sum()), "status", int(ls.status().max()), flush=True)
