#!/usr/bin/env python3
"""Soak test of the periodic FIR kernels: N launches of a 16-stream 2-channel batch, every launch
compared bit for bit with the first one (dynamic scheduling must never change a result or hang).
usage: python tools/soak_fir.py [launches]   (run under `timeout`)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import resampler_amd as ra
from resampler_amd import synth

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
big = len(sys.argv) > 2 and sys.argv[2] == "big"   # the bench workload: 64 streams x 2^20 frames
dev = torch.device("cuda:0")
n_streams, frames = (64, 1 << 20) if big else (16, 150001)
gs = [ra.ResamplerFir.new_from_hz(2, 44100, 48000, ra.Latency.Sample64, ra.Attenuation.Db90) for _ in range(n_streams)]
xs = [synth.fast_noise(2 * frames, seed=700 + (i % 4)) for i in range(n_streams)]
d_in = [torch.from_numpy(x).to(dev) for x in xs]
d_out = [torch.zeros(gs[0].bulk_output_bound(2 * frames, 512), device=dev) for _ in range(n_streams)]
batch = ra.FirBatch(gs)
batch.bind(d_in, d_out)
stream = ra.torch_stream()
first = None
t0 = time.time()
for launch in range(launches):
    batch.reset()
    consumed, produced = batch.resample_bulk_device(512, stream)
    if launch % (50 if big else 8) == 0 or launch < 4:
        torch.cuda.synchronize()
        got = torch.cat([d_out[i][: produced[i]] for i in range(n_streams)])
        if first is None:
            first = got.clone()
        elif not torch.equal(got, first):
            print("MISMATCH at launch", launch)
            sys.exit(1)
torch.cuda.synchronize()
print("soak ok: %d launches, variant %d, %.1f s" % (launches, gs[0].kernel_variant(), time.time() - t0))
