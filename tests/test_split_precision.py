"""CPU check of the arithmetic the split-bf16 FIR kernel relies on (resampler_amd/csrc/fir_split.hip,
DESIGN.md 4.1): an f32 value is exactly the sum of three bf16 values obtained by truncation, and six of the
nine cross products (each exact in f32, accumulated in f32 per 32-tap block like v_mfma_f32_16x16x32_bf16) are
as close to the f64 sum as an f32 FMA chain -- while three products are not enough for the 1e-6 RMS gate."""
import numpy as np


def split3(v):
    u = v.view(np.uint32)
    p1 = (u & np.uint32(0xFFFF0000)).view(np.float32)
    r1 = v - p1
    p2 = (r1.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    r2 = r1 - p2
    p3 = (r2.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    return p1, p2, p3, r2


def test_three_way_split_is_exact():
    rng = np.random.default_rng(7)
    v = np.concatenate([rng.standard_normal(100000).astype(np.float32) * s for s in (1e-20, 1e-3, 1.0, 1e12)])
    p1, p2, p3, r2 = split3(v)
    assert np.array_equal(r2, p3)                                   # nothing is left after three planes
    assert np.array_equal(p1.astype(np.float64) + p2 + p3, v.astype(np.float64))
    for p in (p1, p2, p3):                                          # every plane is a bf16 value
        assert not np.any(p.view(np.uint32) & np.uint32(0xFFFF))


def blockwise(pairs, idx, taps, n):
    acc = np.zeros(n, np.float32)
    for s in range(0, taps, 32):
        for h, x in pairs:                                          # each product exact in f32, summed wide, rounded once
            d = x[idx[:, s:s + 32]].astype(np.float64) @ h[s:s + 32].astype(np.float64)
            acc = (acc.astype(np.float64) + d).astype(np.float32)
    return acc


def test_six_products_match_an_f32_chain():
    rng = np.random.default_rng(1)
    taps, n = 128, 4000
    k = np.arange(taps) - 63.3
    h = (np.sinc(k * 0.9) * np.kaiser(taps, 10)).astype(np.float32)
    x = (rng.standard_normal(n + taps) * 0.3).astype(np.float32)
    idx = np.arange(n)[:, None] + np.arange(taps)[None, :]
    ref = x[idx].astype(np.float64) @ h.astype(np.float64)
    chain = np.zeros(n, np.float32)
    for t in range(taps):
        chain = np.float32(chain + x[idx[:, t]] * h[t])
    h1, h2, h3, _ = split3(h)
    x1, x2, x3, _ = split3(x)
    six = blockwise([(h1, x3), (h2, x2), (h3, x1), (h1, x2), (h2, x1), (h1, x1)], idx, taps, n)
    three = blockwise([(h1, x2), (h2, x1), (h1, x1)], idx, taps, n)
    rms = lambda e: float(np.sqrt(np.mean(e ** 2)))
    assert rms(six - ref) <= 1.0e-7 and rms(six - ref) <= 1.5 * rms(chain - ref)
    assert rms(three - ref) > 1.0e-6                                # bf16x2 would fail the north-star tolerance


def split_f16(v):
    a = v.astype(np.float16).astype(np.float32)
    b = (v - a).astype(np.float32).astype(np.float16).astype(np.float32)
    return a, b


def block_scale(x):
    """The kernel's block floating point (fir_split.hip, `peak`): the power of two that puts the largest magnitude
    of an item's samples into [2^14, 2^15); peaks below 2^-96 are treated as 2^-96."""
    m = float(np.max(np.abs(x)))
    e = max(int(np.floor(np.log2(m))) if m > 0 else -127, -96)
    e = min(e, 11)   # (kPeakMax: peaks of 2^11 and above are not scaled for -- samples of 2^13 and above overflow and are redone in f32)
    return np.float32(2.0 ** (14 - e))


def test_two_fp16_planes_with_three_products_stay_inside_the_gate():
    # the default split kernel: samples scaled by the item's power of two (block floating point), taps by 2^13,
    # two fp16 planes each (round to nearest), products c1 x2 + c2 x1 + c1 x1 accumulated in f32 per 32-tap block
    # like v_mfma_f32_16x16x32_f16.  At EVERY level -- full scale, -60 dB, 2^-24 (the C2 sweep scaled down to one
    # f32 ulp of full scale), 1e-30, samples in the thousands -- the result is as close to the f64 sum, relative to the
    # signal, as the reference's own f32 FMA chain.
    rng = np.random.default_rng(2)
    taps, n = 128, 4000
    k = np.arange(taps) - 63.3
    h = (np.sinc(k * 0.9) * np.kaiser(taps, 10)).astype(np.float32)
    h = (h / h.sum()).astype(np.float32)
    idx = np.arange(n)[:, None] + np.arange(taps)[None, :]
    rms = lambda e: float(np.sqrt(np.mean(e ** 2)))
    h1, h2 = split_f16((h * np.float32(8192.0)).astype(np.float32))
    for scale in (0.99, 0.3, 1e-3, 2.0 ** -17, 2.0 ** -24, 1e-30, 300.0, 4000.0):
        x = (np.clip(rng.standard_normal(n + taps) * 0.4, -1, 1) * scale).astype(np.float32)
        ref = x[idx].astype(np.float64) @ h.astype(np.float64)
        xs = block_scale(x)
        x1, x2 = split_f16((x * xs).astype(np.float32))
        assert np.all(np.isfinite(x1)) and np.abs(x1).max() < 32768.0
        got = blockwise([(h1, x2), (h2, x1), (h1, x1)], idx, taps, n) * np.float32(1.0 / (float(xs) * 8192.0))
        chain = np.zeros(n, np.float32)
        for t in range(taps):
            chain = np.float32(chain + x[idx[:, t]] * h[t])
        assert rms(got - ref) <= 2.5e-7 * rms(ref)                              # relative: far inside 1e-6 of the signal
        assert rms(got - ref) <= 1.05 * rms(chain - ref)                        # at least as close as an f32 FMA chain
    # a loud passage next to a quiet one inside one item: the quiet part keeps 22 bits relative to the item's peak
    x = (np.clip(rng.standard_normal(n + taps) * 0.4, -1, 1)).astype(np.float32)
    x[: n // 2] *= np.float32(1e-5)
    ref = x[idx].astype(np.float64) @ h.astype(np.float64)
    xs = block_scale(x)
    x1, x2 = split_f16((x * xs).astype(np.float32))
    got = blockwise([(h1, x2), (h2, x1), (h1, x1)], idx, taps, n) * np.float32(1.0 / (float(xs) * 8192.0))
    quiet = slice(0, n // 2 - taps)
    assert rms((got - ref)[quiet]) <= 2e-7 * float(np.abs(x).max())             # absolute error set by the item's peak
    assert rms(got - ref) <= 1e-7
