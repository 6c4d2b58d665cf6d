"""Synthetic inputs for tests and bench.

* ``sweep`` restates the log sine sweep of the reference's quality harness
  (test_audio_resampler.py:75-96: 20 Hz -> 0.95 * fs/2 logarithmic chirp, 0.1 s linear fades,
  x 0.99) in closed form, with a per-channel phase offset of c*pi/7 so channel mix-ups show.
* ``lcg_noise`` restates the white-noise generator of the reference's criterion benches
  (benches/benchmark_resampler_fir.rs:12-21).
"""
from __future__ import annotations

import numpy as np


def sweep(n_frames: int, channels: int, fs: float, f0: float = 20.0, f1_frac: float = 0.95,
          amplitude: float = 0.99) -> np.ndarray:
    """Interleaved f32 [n_frames * channels]."""
    duration = n_frames / fs
    t = np.linspace(0.0, duration, n_frames)
    f1 = fs / 2.0 * f1_frac
    k = f1 / f0
    # scipy.signal.chirp(method='logarithmic'): phase = 2*pi*f0*T/ln(k) * (k**(t/T) - 1)
    phase = 2.0 * np.pi * f0 * duration / np.log(k) * (np.power(k, t / duration) - 1.0)
    fade = np.ones(n_frames)
    fade_samples = min(int(0.1 * fs), n_frames // 2)
    if fade_samples > 0:
        fade[:fade_samples] = np.linspace(0.0, 1.0, fade_samples)
        fade[-fade_samples:] = np.linspace(1.0, 0.0, fade_samples)
    out = np.empty((n_frames, channels), np.float32)
    for c in range(channels):
        out[:, c] = (amplitude * fade * np.sin(phase + c * np.pi / 7.0)).astype(np.float32)
    return out.reshape(-1)


def lcg_noise(n_values: int, seed_offset: int = 0) -> np.ndarray:
    """benches/benchmark_resampler_fir.rs:12-21 (u128 multiplicative LCG, top 64 bits)."""
    state = (456423156461231 + seed_offset) & ((1 << 128) - 1)
    mul = 0xDA942042E4DD58B5
    mask = (1 << 128) - 1
    u64max = float((1 << 64) - 1)
    out = np.empty(n_values, np.float32)
    for i in range(n_values):
        state = (state * mul) & mask
        val = state >> 64
        out[i] = np.float32(np.float32(val / u64max) * np.float32(2.0) - np.float32(1.0))
    return out


def fast_noise(n_values: int, seed: int = 0) -> np.ndarray:
    """Uniform [-1, 1) f32 noise from numpy's PCG64 (for large buffers)."""
    rng = np.random.default_rng(seed)
    return (rng.random(n_values, dtype=np.float32) * np.float32(2.0) - np.float32(1.0))


def hash_noise(n_values: int, seed: int = 0) -> np.ndarray:
    """Uniform [-1, 1) f32 noise from a counter-based generator (splitmix64 of seed + index, top 24 bits):
    pure integer arithmetic, so the committed golden fixtures do not depend on a numpy version."""
    with np.errstate(over="ignore"):
        z = (np.arange(n_values, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
             + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    top = (z >> np.uint64(40)).astype(np.float64)          # 24 bits
    return (top / float(1 << 23) - 1.0).astype(np.float32)
