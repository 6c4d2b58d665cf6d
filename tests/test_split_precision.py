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


def test_two_fp16_planes_with_three_products_stay_inside_the_gate():
    # the default split kernel: operands scaled by 2^12 / 2^13, two fp16 planes each (round to nearest),
    # products c1 x2 + c2 x1 + c1 x1 accumulated in f32 per 32-tap block like v_mfma_f32_16x16x32_f16
    rng = np.random.default_rng(2)
    taps, n = 128, 4000
    k = np.arange(taps) - 63.3
    h = (np.sinc(k * 0.9) * np.kaiser(taps, 10)).astype(np.float32)
    h = (h / h.sum()).astype(np.float32)
    idx = np.arange(n)[:, None] + np.arange(taps)[None, :]
    rms = lambda e: float(np.sqrt(np.mean(e ** 2)))
    for scale in (0.99, 0.3, 1e-3, 1e-6):
        x = np.clip(rng.standard_normal(n + taps) * scale, -1, 1).astype(np.float32)
        ref = x[idx].astype(np.float64) @ h.astype(np.float64)
        h1, h2 = split_f16((h * np.float32(8192.0)).astype(np.float32))
        x1, x2 = split_f16((x * np.float32(4096.0)).astype(np.float32))
        got = blockwise([(h1, x2), (h2, x1), (h1, x1)], idx, taps, n) * np.float32(1.0 / (4096.0 * 8192.0))
        chain = np.zeros(n, np.float32)
        for t in range(taps):
            chain = np.float32(chain + x[idx[:, t]] * h[t])
        assert rms(got - ref) <= 1.0e-7                                         # far inside the 1e-6 gate (6e-8 at full scale)
        if scale >= 1e-3:
            assert rms(got - ref) <= rms(chain - ref)                           # at least as close as an f32 FMA chain
        assert rms(got - ref) <= 1e-5 * rms(ref)                                # quiet signals keep their relative precision (-100 dB)
    # samples of magnitude >= 16 overflow the scaled fp16 plane: the kernel's non-finite check takes over
    assert not np.isfinite(np.float16(np.float32(16.0) * np.float32(4096.0)))
