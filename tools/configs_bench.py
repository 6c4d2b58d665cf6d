#!/usr/bin/env python3
"""Secondary measurements for BASELINE configs 4 and 5 on ONE GPU (the multi-GPU form shards
streams with resampler_amd.sharding and runs this per rank).

  c4: batch of N independent ResamplerFir streams, stream i = ordered pair i mod 6 of the
      44.1k / 48k / 96k conversions, 2 ch, 128 taps, steps of 512 frames per stream, one launch
      per step (rsmp_fir_batch_resample_bulk_device).
  c5: ResamplerFir 8 ch 96000 -> 44100, 128 taps, Db120, 512-frame chunks: per-chunk latency of
      the drop-in host call (H2D + launch + D2H) and of the device-resident call.

Prints one JSON object per config.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def c4(n_streams: int, steps: int, frames: int):
    import torch
    import resampler_amd as ra
    from resampler_amd import sharding, synth

    dev = torch.device("cuda:0")
    specs = sharding.mixed_rate_batch(n_streams, 2, frames)
    hs = [ra.ResamplerFir.new_from_hz(s.channels, s.in_hz, s.out_hz, ra.Latency.Sample64,
                                      ra.Attenuation.Db90) for s in specs]
    x = torch.from_numpy(synth.fast_noise(2 * frames, seed=1)).to(dev)
    d_in = [x.clone() for _ in specs]
    d_out = [torch.empty(h.bulk_output_bound(2 * frames, 2 * frames), device=dev) for h in hs]
    batch = ra.FirBatch(hs)
    batch.bind(d_in, d_out)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        consumed, produced = batch.resample_bulk_device(2 * frames, stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tot_out = 0
    for _ in range(steps):
        consumed, produced = batch.resample_bulk_device(2 * frames, stream)
        tot_out += sum(produced)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    vals_in = n_streams * 2 * frames * steps
    return {"config": "c4", "streams": n_streams, "frames_per_step": frames, "steps": steps,
            "ms_per_step": round(dt / steps * 1e3, 4), "host_ms_per_step": round(host / steps * 1e3, 4),
            "Msamples_in_per_s": round(vals_in / dt / 1e6, 1),
            "GBps_algorithmic": round(4.0 * (vals_in + tot_out) / dt / 1e9, 2)}


def c5(chunks: int):
    import torch
    import resampler_amd as ra
    from resampler_amd import synth

    dev = torch.device("cuda:0")
    ch, frames = 8, 512
    h = ra.ResamplerFir.new(ch, ra.SampleRate.Hz96000, ra.SampleRate.Hz44100, ra.Latency.Sample64,
                            ra.Attenuation.Db120)
    x = synth.fast_noise(ch * frames, seed=2)
    out = np.zeros(h.buffer_size_output(), np.float32)
    lat = []
    for i in range(chunks + 20):
        t0 = time.perf_counter()
        h.resample(x, out)
        lat.append(time.perf_counter() - t0)
    lat = np.array(lat[20:]) * 1e6
    d_x = torch.from_numpy(x).to(dev)
    d_y = torch.empty(h.buffer_size_output(), device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    lat_d = []
    for i in range(chunks + 20):
        t0 = time.perf_counter()
        h.resample_device(d_x, d_y, stream)
        torch.cuda.synchronize()
        lat_d.append(time.perf_counter() - t0)
    lat_d = np.array(lat_d[20:]) * 1e6
    return {"config": "c5", "channels": ch, "chunk_frames": frames, "chunks": chunks,
            "host_call_us": {"p50": round(float(np.percentile(lat, 50)), 1),
                             "p99": round(float(np.percentile(lat, 99)), 1)},
            "device_call_us": {"p50": round(float(np.percentile(lat_d, 50)), 1),
                               "p99": round(float(np.percentile(lat_d, 99)), 1)},
            "realtime_factor_p50": round(frames / 96000.0 / (float(np.percentile(lat, 50)) * 1e-6), 1)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--chunks", type=int, default=2000)
    a = ap.parse_args()
    print(json.dumps(c4(a.streams, a.steps, a.frames)), flush=True)
    print(json.dumps(c5(a.chunks)), flush=True)
