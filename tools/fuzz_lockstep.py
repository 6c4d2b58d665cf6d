#!/usr/bin/env python3
"""Randomised parity soak of the lock-step batch against the CPU oracle (GPU box): random rate pairs
(rational and not), channel counts, tap counts, step sizes (also ragged per stream), stream histories.
usage: python tools/fuzz_lockstep.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import resampler_amd as ra
from oracle import pyoracle as o

RATES = [8000, 11025, 16000, 22050, 32000, 44100, 48000, 88200, 96000, 176400, 192000, 44101, 47999, 12345]
LAT = [ra.Latency.Sample8, ra.Latency.Sample16, ra.Latency.Sample32, ra.Latency.Sample64]
ATT = {ra.Attenuation.Db60: 60, ra.Attenuation.Db90: 90, ra.Attenuation.Db120: 120}

def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    t0 = time.time()
    rounds = streams_total = 0
    worst = 0.0
    while time.time() - t0 < budget:
        n = int(rng.integers(1, 40))
        frames = int(rng.choice([1, 7, 64, 200, 512, 1000, 2048]))
        steps = int(rng.integers(2, 7))
        lat = LAT[int(rng.integers(0, 4))]
        att = list(ATT)[int(rng.integers(0, 3))]
        n_pairs = int(rng.integers(1, 5))
        pairs = [(int(rng.choice(RATES)), int(rng.choice(RATES))) for _ in range(n_pairs)]
        chans = [int(rng.integers(1, 5)) for _ in range(n_pairs)]
        specs = [(chans[i % n_pairs],) + pairs[i % n_pairs] for i in range(n)]
        if os.environ.get("RSMP_FUZZ_VERBOSE"):   # (a GPU fault kills the process: the last line names the round)
            print("round", rounds, "n", n, "frames", frames, "steps", steps, lat, att, "pairs", pairs, "channels", chans, flush=True)
        hs, refs = [], []
        try:
            for ch, a, b in specs:
                hs.append(ra.ResamplerFir.new_from_hz(ch, a, b, lat, att))
                refs.append(o.OracleFir(ch, a, b, lat.taps(), ATT[att], o.CONVOLVE_AVX_FMA if o.have_avx_fma() else o.CONVOLVE_SCALAR))
        except Exception as e:
            print("skip", specs[0], e)
            continue
        # histories: some streams have already run (at the level they will go on with)
        base_levels = [2.0 ** float(rng.integers(-30, 7)) for _ in specs]
        for h, r, (ch, a, b), blv in zip(hs, refs, specs, base_levels):
            pre = int(rng.integers(0, 3)) * int(rng.integers(1, 700))
            if pre:
                x = ((rng.random(pre * ch, dtype=np.float32) * 2 - 1).astype(np.float32) * np.float32(blv)).astype(np.float32)
                og = np.zeros(h.buffer_size_output(), np.float32); orr = np.zeros(r.buffer_size_output(), np.float32)
                off = 0
                while off < x.size:
                    cg, pg = h.resample(x[off:off + 256 * ch], og)
                    rc, cr, pr = r.resample(x[off:off + 256 * ch], orr)
                    assert rc == 0 and (cg, pg) == (cr, pr), ("prefeed", specs, (cg, pg), (cr, pr))
                    off += cg
                    if cg == 0: break
        ragged = rng.random() < 0.4
        fr = [int(rng.integers(0, frames + 1)) for _ in range(n)] if ragged else None
        # every stream at a level of its own (2^-30 .. 2^6), some changing level from one step to the next
        xs, levels = [], []
        for (ch, _, _), lv in zip(specs, base_levels):
            x = (rng.random(steps * frames * ch, dtype=np.float32) * 2 - 1).astype(np.float32) * np.float32(lv)
            if rng.integers(3) == 0:
                k = int(rng.integers(0, steps)) * frames * ch
                j = 2.0 ** float(rng.integers(-16, 6))
                x[k:] *= np.float32(j)
                lv = max(lv, lv * j)
            xs.append(x.astype(np.float32))
            levels.append(lv)
        caps = [h.buffer_size_output() for h in hs]
        d_in = [torch.from_numpy(x).to(dev) for x in xs]
        d_out = [torch.zeros(steps * c, device=dev) for c in caps]
        try:
            ls = ra.FirLockstep(hs, frames)
        except ra.ResampleError as e:
            print("lockstep refused:", e)
            continue
        ls.bind_caps(d_in, d_out, caps)
        d_fr = torch.tensor(fr, dtype=torch.int32, device=dev) if ragged else None
        want = [[] for _ in range(n)]
        orr = [np.zeros(c, np.float32) for c in caps]
        for k in range(steps):
            ls.step(frames, k * frames, append=True, d_in_frames=d_fr)
            cons, prod = ls.counts()
            for i, (ch, a, b) in enumerate(specs):
                f = frames if not ragged else fr[i]
                rc, cr, pr = refs[i].resample(xs[i][k * frames * ch:(k * frames + f) * ch], orr[i])
                assert rc == 0
                assert (int(cons[i]), int(prod[i])) == (cr, pr), ("counts", specs[i], lat, frames, k, (cons[i], prod[i]), (cr, pr))
                want[i].append(orr[i][:pr].copy())
        for i in range(n):
            w = np.concatenate(want[i]) if want[i] else np.zeros(0, np.float32)
            g = d_out[i][:w.size].cpu().numpy()
            # the north-star gate (1e-6 RMS on full-scale audio) relative to the stream's own full scale
            e = float(np.sqrt(np.mean((g.astype(np.float64) - w) ** 2))) / levels[i] if w.size else 0.0
            worst = max(worst, e)
            if e > 1e-6:
                d = np.abs(g.astype(np.float64) - w)
                k = int(np.argmax(d))
                print("FAIL stream", i, specs[i], lat, att, "frames", frames, "steps", steps, "ragged", ragged, "level", levels[i], "n_out", w.size,
                      "max abs err", d[k], "at", k, "got", g[max(0, k - 2):k + 3], "want", w[max(0, k - 2):k + 3], "bad count", int(np.sum(d > 1e-5 * levels[i])),
                      "per-step outputs", [len(v) for v in want[i]], "status", ls.status()[i] if hasattr(ls, "status") else None, flush=True)
            assert e <= 1e-6, ("rms", specs[i], lat, att, frames, e)
        ls.sync()
        for h, r in zip(hs, refs):
            assert h.state() == r.state(), ("state", h.state(), r.state())
        st = ls.status()
        rounds += 1
        streams_total += n
        ls.close()
    print(f"fuzz_lockstep: {rounds} rounds, {streams_total} streams, worst rms {worst:.3e}, seed {seed}: OK")

if __name__ == "__main__":
    main()
