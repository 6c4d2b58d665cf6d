#!/usr/bin/env python3
"""Secondary measurement for BASELINE config 5 on ONE GPU (config 4 is `bench.py --config c4`).

  c5: ResamplerFir 8 ch 96000 -> 44100, 128 taps, Db120, 512-frame chunks: per-chunk latency of
      the drop-in host call (H2D + launch + D2H) and of the device-resident call, and the bulk rate
      of the same stream (10 minutes = 57.6 M frames fed as 2^20-frame bulk calls).

Prints one JSON object.
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
    # the 10-minute stream in bulk: 55 calls of 2^20 frames (57.6 M frames), HBM resident
    n_bulk = 1 << 20
    xb = torch.from_numpy(synth.fast_noise(ch * n_bulk, seed=3)).to(dev)
    yb = torch.empty(h.bulk_output_bound(ch * n_bulk, ch * frames), device=dev)
    h.resample_bulk_device(xb, yb, ch * frames, stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    calls = 55
    out_values = 0
    for _ in range(calls):
        c_, p_ = h.resample_bulk_device(xb, yb, ch * frames, stream)
        out_values += p_
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bulk = {"seconds_for_10_minutes_of_audio": round(dt, 4), "Msamples_in_per_s": round(calls * ch * n_bulk / dt / 1e6, 1),
            "GBps_algorithmic": round(4.0 * (calls * ch * n_bulk + out_values) / dt / 1e9, 1),
            "kernel_variant": h.kernel_variant()}
    return {"config": "c5", "channels": ch, "chunk_frames": frames, "chunks": chunks, "bulk_10_min": bulk,
            "host_call_us": {"p50": round(float(np.percentile(lat, 50)), 1),
                             "p99": round(float(np.percentile(lat, 99)), 1)},
            "device_call_us": {"p50": round(float(np.percentile(lat_d, 50)), 1),
                               "p99": round(float(np.percentile(lat_d, 99)), 1)},
            "realtime_factor_p50": round(frames / 96000.0 / (float(np.percentile(lat, 50)) * 1e-6), 1)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=2000)
    a = ap.parse_args()
    print(json.dumps(c5(a.chunks)), flush=True)
