#!/usr/bin/env python3
"""Randomised parity soak of the bulk / batch FIR launches (periodic kernels of every flavour, generic kernel)
and the FFT bulk path against the CPU oracle (GPU box): random rate pairs, channel counts, tap counts,
attenuations, stream lengths, call sizes, kernel modes, stream histories.
usage: python tools/fuzz_bulk.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import resampler_amd as ra
from oracle import pyoracle as o

RATES = [8000, 11025, 16000, 22050, 32000, 44100, 48000, 88200, 96000, 176400, 192000, 44101, 12345]
FFT_RATES = [22050, 16000, 32000, 44100, 48000, 88200, 96000, 176400, 192000, 384000]
LAT = [ra.Latency.Sample8, ra.Latency.Sample16, ra.Latency.Sample32, ra.Latency.Sample64]
ATT = {ra.Attenuation.Db60: 60, ra.Attenuation.Db90: 90, ra.Attenuation.Db120: 120}
MODES = [ra.FirKernel.Auto, ra.FirKernel.Auto, ra.FirKernel.Periodic, ra.FirKernel.PeriodicVector, ra.FirKernel.PeriodicF32,
         ra.FirKernel.Generic]


def rms(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))) if a.size else 0.0


def fir_round(rng, dev):
    ch = int(rng.choice([1, 2, 2, 2, 3, 4, 6, 8]))
    a, b = int(rng.choice(RATES)), int(rng.choice(RATES))
    lat = LAT[int(rng.integers(0, 4))]
    att = list(ATT)[int(rng.integers(0, 3))]
    mode = MODES[int(rng.integers(0, len(MODES)))]
    n = int(rng.integers(1, 5))
    frames = int(rng.choice([3000, 20000, 60000, 150000]))
    chunk = int(rng.choice([64, 500, 512, 1024, 4096])) * ch
    desc = ("fir", ch, a, b, lat, att, mode, n, frames, chunk)
    if os.environ.get("RSMP_FUZZ_VERBOSE"):
        print(desc, flush=True)
    hs = [ra.ResamplerFir.new_from_hz(ch, a, b, lat, att) for _ in range(n)]
    refs = [o.OracleFir(ch, a, b, lat.taps(), ATT[att]) for _ in range(n)]
    for h in hs:
        h.set_kernel(mode)
    # histories
    for h, r in zip(hs, refs):
        pre = int(rng.integers(0, 2)) * int(rng.integers(1, 900))
        if pre:
            x = (rng.random(pre * ch, dtype=np.float32) * 2 - 1).astype(np.float32)
            og, orr = np.zeros(h.buffer_size_output(), np.float32), np.zeros(r.buffer_size_output(), np.float32)
            off = 0
            while off < x.size:
                cg, pg = h.resample(x[off:off + 256 * ch], og)
                rc, cr, pr = r.resample(x[off:off + 256 * ch], orr)
                assert rc == 0 and (cg, pg) == (cr, pr), ("prefeed", desc)
                assert rms(og[:pg], orr[:pr]) <= 1e-6, ("prefeed rms", desc)
                off += cg
                if cg == 0:
                    break
    xs = [(rng.random(frames * ch, dtype=np.float32) * 2 - 1).astype(np.float32) for _ in range(n)]
    d_in = [torch.from_numpy(x).to(dev) for x in xs]
    d_out = [torch.zeros(h.bulk_output_bound(x.size, chunk), device=dev) for h, x in zip(hs, xs)]
    batch = ra.FirBatch(hs)
    batch.bind(d_in, d_out)
    cons, prod = batch.resample_bulk_device(chunk, ra.torch_stream())
    torch.cuda.synchronize()
    worst = 0.0
    for i in range(n):
        y, calls = refs[i].resample_all(xs[i], chunk)
        assert int(prod[i]) == y.size and int(cons[i]) == int(calls[:, 0].sum()), ("counts", desc, i)
        e = rms(d_out[i][:y.size].cpu().numpy(), y)
        assert e <= 1e-6, ("rms", desc, i, e, hs[i].kernel_variant())
        assert hs[i].state() == refs[i].state(), ("state", desc, i)
        worst = max(worst, e)
    return worst


def fft_round(rng, dev):
    ch = int(rng.choice([1, 2, 2, 3]))
    a, b = int(rng.choice(FFT_RATES)), int(rng.choice(FFT_RATES))
    if a == b:
        return 0.0
    blocks = int(rng.integers(1, 40))
    n = int(rng.integers(1, 4))
    desc = ("fft", ch, a, b, blocks, n)
    if os.environ.get("RSMP_FUZZ_VERBOSE"):
        print(desc, flush=True)
    sr = lambda hz: ra.SampleRate(FFT_RATES.index(hz))
    # parity is claimed where the reference's per-channel scratch regions do not collide
    fi, fo, _, _ = o.fft_plan(a, b)
    stride = fi * (ch + 1)
    if not (ch == 1 or (fo <= stride and (ch - 1) * stride + fo <= fo * (ch + 1) * ch)):
        return 0.0
    refs = [o.OracleFft(ch, a, b) for _ in range(n)]
    hs = [ra.ResamplerFft.new(ch, sr(a), sr(b)) for _ in range(n)]
    n_in, n_out = hs[0].chunk_size_input(), hs[0].chunk_size_output()
    xs = [(rng.random(blocks * n_in, dtype=np.float32) * 2 - 1).astype(np.float32) for _ in range(n)]
    d_in = [torch.from_numpy(x).to(dev) for x in xs]
    d_out = [torch.zeros(blocks * n_out, device=dev) for _ in range(n)]
    batch = ra.FftBatch(hs)
    batch.bind(d_in, d_out, [blocks] * n)
    batch.resample_bulk_device(ra.torch_stream())
    torch.cuda.synchronize()
    worst = 0.0
    for i in range(n):
        ref = np.zeros((blocks, n_out), np.float32)
        for k in range(blocks):
            assert refs[i].resample(xs[i][k * n_in:(k + 1) * n_in], ref[k]) == 0, ("oracle", desc)
        e = rms(d_out[i].cpu().numpy(), ref.reshape(-1))
        assert e <= 1e-6, ("rms", desc, i, e)
        worst = max(worst, e)
    return worst


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    t0 = time.time()
    rounds = 0
    worst = 0.0
    while time.time() - t0 < budget:
        worst = max(worst, fir_round(rng, dev) if rng.random() < 0.75 else fft_round(rng, dev))
        rounds += 1
    print(f"fuzz_bulk: {rounds} rounds, worst rms {worst:.3e}, seed {seed}: OK")


if __name__ == "__main__":
    main()
