#!/usr/bin/env python3
"""Generates tests/golden/config_fixtures.json from the CPU oracle (oracle/, the C restatement of the
reference) in the BUILD container.  SURVEY 8(c): the reference's own tests pin no end-to-end sample value
or (consumed, produced) sequence, so the oracle's outputs for the five BASELINE configs are frozen here --
first / last values, every count, a SHA-256 of the full stream, plus hashes of the polyphase tables and of
the FFT filter spectrum.  tests/test_golden_configs.py checks the oracle (any box: a different libm moves
the tables and shows up here) and the HIP path (GPU box) against them.

    python tests/golden/make_fixtures.py        # rewrites config_fixtures.json
"""
import base64
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import pyoracle as o          # noqa: E402
from resampler_amd import sharding, synth  # noqa: E402

KEEP = 4096


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a, np.float32).tobytes()).hexdigest()


def pack(a: np.ndarray) -> str:
    """little-endian f32 bytes, base64 (exact values, a quarter of the size of a decimal list)"""
    return base64.b64encode(np.ascontiguousarray(a, "<f4").tobytes()).decode()


def head_tail(y: np.ndarray, keep: int = KEEP):
    return pack(y[:keep]), pack(y[-keep:])


def rle(calls: np.ndarray):
    """[(consumed, produced, repeat), ...] of a calls[n, 2] array."""
    out = []
    for c, p in calls.tolist():
        if out and out[-1][0] == c and out[-1][1] == p:
            out[-1][2] += 1
        else:
            out.append([c, p, 1])
    return out


def fir_config(name, ch, in_hz, out_hz, att_db, x, chunk, kind=o.CONVOLVE_SCALAR):
    r = o.OracleFir(ch, in_hz, out_hz, 128, att_db, kind)
    y, calls = r.resample_all(x, chunk)
    h, t = head_tail(y)
    return {"name": name, "channels": ch, "in_hz": in_hz, "out_hz": out_hz, "taps": 128, "attenuation_db": att_db,
            "chunk_values": chunk, "in_values": int(x.size), "out_values": int(y.size), "calls_rle": rle(calls),
            "sha256": sha(y), "head": h, "tail": t, "table_sha256": sha(r.coeffs()),
            "buffer_size_output": r.buffer_size_output(), "final_state": list(r.state())}


C5_FULL_FRAMES = 57_600_000   # BASELINE config 5 as stated: ten minutes of 8-channel audio at 96 kHz


def c5_full_input() -> np.ndarray:
    """The ten-minute stream of config 5: hash_noise(seed 55), generated in pieces (the generator's 64-bit temporaries of
    460.8 M values at once would be ~11 GB)."""
    n = C5_FULL_FRAMES * 8
    x = np.empty(n, np.float32)
    step = 1 << 24
    for a in range(0, n, step):
        b = min(n, a + step)
        with np.errstate(over="ignore"):
            z = (np.arange(a, b, dtype=np.uint64) + np.uint64(55) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0x9E3779B97F4A7C15))
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        x[a:b] = ((z >> np.uint64(40)).astype(np.float64) / float(1 << 23) - 1.0).astype(np.float32)
    return x


def c5_full():
    """Config 5 at its stated length through convolve_interp_avx_fma, 512-frame calls: hashes only (0.85 GB of output)."""
    r = o.OracleFir(8, 96000, 44100, 128, 120, o.CONVOLVE_AVX_FMA)
    x = c5_full_input()
    assert np.array_equal(x[:4096], synth.hash_noise(4096, seed=55))
    y, calls = r.resample_all(x, 512 * 8)
    return {"name": "c5_full", "channels": 8, "in_hz": 96000, "out_hz": 44100, "taps": 128, "attenuation_db": 120,
            "chunk_values": 512 * 8, "in_values": int(x.size), "out_values": int(y.size), "n_calls": int(calls.shape[0]),
            "calls_sha256": hashlib.sha256(np.ascontiguousarray(calls, "<i8").tobytes()).hexdigest(),
            "sha256": sha(y), "final_state": list(r.state())}


def main():
    fx = {"generator": "tests/golden/make_fixtures.py", "oracle": "oracle/*.c (c1 .. c5: scalar convolve; *_avx_fma and c4_256: convolve_interp_avx_fma)"}
    # C1: 1 ch 48000 -> 44100, Sample64 / Db90, 512-sample calls, 2^20-frame sweep
    fx["c1"] = fir_config("c1", 1, 48000, 44100, 90, synth.sweep(1 << 20, 1, 48000.0), 512)
    # C2: 2 ch 44100 -> 48000, 128 taps, 2^20-frame sweep, the CLI's 512-value calls
    fx["c2"] = fir_config("c2", 2, 44100, 48000, 90, synth.sweep(1 << 20, 2, 44100.0), 512)
    # C3: ResamplerFft 2 ch 44100 -> 48000, 892 blocks of 1176 frames
    f = o.OracleFft(2, 44100, 48000)
    n_in, n_out = f.chunk_size_input(), f.chunk_size_output()
    x = synth.sweep(892 * n_in // 2, 2, 44100.0)
    y = np.zeros(892 * n_out, np.float32)
    blk = np.zeros(n_out, np.float32)
    for b in range(892):
        assert f.resample(x[b * n_in:(b + 1) * n_in], blk) == 0
        y[b * n_out:(b + 1) * n_out] = blk
    h, t = head_tail(y)
    fx["c3"] = {"name": "c3", "blocks": 892, "chunk_size_input": n_in, "chunk_size_output": n_out, "sha256": sha(y),
                "head": h, "tail": t, "filter_spectrum_sha256": sha(f.filter_spectrum().view(np.float32))}
    # C4: 1024 mixed-rate streams, 16 lock-step steps of 512 frames, stream i fed hash_noise(seed = i)
    specs = sharding.mixed_rate_batch(1024, 2, 512)
    steps = 16
    streams = []
    all_hash = hashlib.sha256()
    for i, s in enumerate(specs):
        r = o.OracleFir(2, s.in_hz, s.out_hz, 128, 90)
        x = synth.hash_noise(steps * 512 * 2, seed=i)
        out = np.zeros(r.buffer_size_output(), np.float32)
        ys, counts = [], []
        for k in range(steps):
            rc, c, p = r.resample(x[k * 1024:(k + 1) * 1024], out)
            assert rc == 0
            counts.append([c, p])
            ys.append(out[:p].copy())
        y = np.concatenate(ys)
        digest = sha(y)
        all_hash.update(bytes.fromhex(digest))
        if i < 12 or i >= 1018:   # two streams of every pair at both ends are kept in detail
            streams.append({"index": i, "in_hz": s.in_hz, "out_hz": s.out_hz, "counts": counts, "sha256": digest,
                            "head": pack(y[:256]), "tail": pack(y[-256:])})
    fx["c4"] = {"name": "c4", "streams": 1024, "steps": steps, "frames_per_step": 512, "detail": streams,
                "sha256_of_stream_sha256s": all_hash.hexdigest()}
    # C5: 8 ch 96000 -> 44100, 128 taps, Db120, 512-frame chunks (64 of them)
    fx["c5"] = fir_config("c5", 8, 96000, 44100, 120, synth.hash_noise(64 * 512 * 8, seed=5), 512 * 8)
    # C2 / C5 again through convolve_interp_avx_fma (src/fir/avx.rs:5-61) -- the CPU SIMD path north_star names; its
    # lane structure and FMAs are exact operations, so every CPU with AVX + FMA returns these bits
    assert o.have_avx_fma(), "fixtures are generated on a CPU with AVX + FMA"
    fx["c2_avx_fma"] = fir_config("c2_avx_fma", 2, 44100, 48000, 90, synth.sweep(1 << 20, 2, 44100.0), 512, o.CONVOLVE_AVX_FMA)
    fx["c5_avx_fma"] = fir_config("c5_avx_fma", 8, 96000, 44100, 120, synth.hash_noise(64 * 512 * 8, seed=5), 512 * 8,
                                  o.CONVOLVE_AVX_FMA)
    # C4 at its full length (256 steps) for two streams of every rate pair, AVX + FMA path
    long_detail = []
    for i in list(range(12)):
        s = specs[i]
        r = o.OracleFir(2, s.in_hz, s.out_hz, 128, 90, o.CONVOLVE_AVX_FMA)
        x = synth.hash_noise(256 * 512 * 2, seed=i)
        out = np.zeros(r.buffer_size_output(), np.float32)
        ys, counts = [], []
        for k in range(256):
            rc, c, p = r.resample(x[k * 1024:(k + 1) * 1024], out)
            assert rc == 0
            counts.append([c, p])
            ys.append(out[:p].copy())
        y = np.concatenate(ys)
        long_detail.append({"index": i, "in_hz": s.in_hz, "out_hz": s.out_hz, "counts_rle": rle(np.array(counts)),
                            "sha256": sha(y), "head": pack(y[:256]), "tail": pack(y[-256:]), "final_state": list(r.state())})
    fx["c4_256"] = {"name": "c4_256", "steps": 256, "frames_per_step": 512, "detail": long_detail}
    path = os.path.join(ROOT, "tests", "golden", "config_fixtures.json")
    fx["c5_full"] = c5_full()
    with open(path, "w") as fh:
        json.dump(fx, fh)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    if "--c5-full-only" in sys.argv:   # adds / refreshes that one entry of the committed file (the rest takes minutes)
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config_fixtures.json")
        with open(path) as fh:
            fx = json.load(fh)
        fx["c5_full"] = c5_full()
        with open(path, "w") as fh:
            json.dump(fx, fh)
        print("c5_full:", fx["c5_full"])
    else:
        main()
