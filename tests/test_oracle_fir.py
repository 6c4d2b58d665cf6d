"""Pins the CPU oracle's FIR half against every known answer the reference's own tests hold
(src/window.rs:152-410, src/fir/mod.rs:137-247, src/resampler_fir.rs:693-862)."""
import numpy as np
import pytest

from oracle import pyoracle as o


def rel(a, e):
    return abs(a / e - 1.0)


def test_bessel_i0_known_values(golden):        # window.rs:152-160
    for x, expected in golden["bessel_i0"]:
        assert rel(o.bessel_i0(x), expected) < 1e-6


@pytest.mark.parametrize("kind,wt", [("kaiser_periodic", o.WINDOW_PERIODIC),
                                     ("kaiser_symmetric", o.WINDOW_SYMMETRIC)])
def test_kaiser_windows(golden, kind, wt):      # window.rs:162-227, 296-362
    for case in golden[kind]:
        w = o.make_kaiser_window(case["n"], case["beta"], wt)
        assert len(w) == case["n"]
        for a, e in zip(w, case["w"]):
            assert rel(float(a), e) < 1e-5


def test_cutoff_kaiser(golden):                 # window.rs:229-247
    for n, expected in golden["cutoff_kaiser_beta10"]:
        assert rel(o.calculate_cutoff_kaiser(n, 10.0), expected) < 1e-6
    for n in (32, 64, 128, 256, 512, 1024, 2048):
        assert 0.0 < o.calculate_cutoff_kaiser(n, 10.0) < 1.0


def test_make_sincs_reference_values(golden):   # window.rs:273-294, 364-385
    for key, wt in (("sincs_4_2_0p9_beta10_periodic", o.WINDOW_PERIODIC),
                    ("sincs_4_2_0p9_beta10_symmetric", o.WINDOW_SYMMETRIC)):
        got = o.make_sincs_for_kaiser(4, 2, 0.9, 10.0, wt)
        assert got.shape == (2, 4)
        for row, exp_row in zip(got, golden[key]):
            for a, e in zip(row, exp_row):
                assert rel(float(a), e) < 1e-5


def test_make_sincs_normalization():            # window.rs:387-410
    got = o.make_sincs_for_kaiser(8, 4, 0.95, 10.0, o.WINDOW_PERIODIC)
    assert abs(float(got.sum()) - 4.0) < 0.01


@pytest.mark.parametrize("taps", [16, 32, 64, 128])
def test_convolve_avx_matches_scalar(taps):     # fir/mod.rs:137-192
    if not o.have_avx_fma():
        pytest.skip("no AVX+FMA on this host")
    i = np.arange(taps, dtype=np.float32)
    x = np.sin(i * np.float32(0.1)).astype(np.float32)
    c1 = np.cos(i * np.float32(0.05)).astype(np.float32)
    c2 = np.cos(i * np.float32(0.05) + np.float32(0.1)).astype(np.float32)
    for frac in (0.0, 0.25, 0.5, 0.75, 1.0):
        a = o.convolve_interp(x, c1, c2, frac, o.CONVOLVE_AVX_FMA)
        s = o.convolve_interp(x, c1, c2, frac, o.CONVOLVE_SCALAR)
        assert abs(a - s) < 1e-5


@pytest.mark.parametrize("taps", [16, 32, 64, 128])
def test_convolve_avx512_matches_scalar(taps):  # fir/mod.rs:137-192, for fir/avx512.rs:5-50
    """The leaf the reference's runtime dispatch takes FIRST where the CPU has avx512f (resampler_fir.rs:331-345) -- what its
    published Zen 5 figures ran (bench.py cpu_baseline `avx512`).  The reference's own differential recipe and tolerance."""
    if not o.have_avx512f():
        pytest.skip("no AVX-512F on this host")
    i = np.arange(taps, dtype=np.float32)
    x = np.sin(i * np.float32(0.1)).astype(np.float32)
    c1 = np.cos(i * np.float32(0.05)).astype(np.float32)
    c2 = np.cos(i * np.float32(0.05) + np.float32(0.1)).astype(np.float32)
    for frac in (0.0, 0.25, 0.5, 0.75, 1.0):
        a = o.convolve_interp(x, c1, c2, frac, o.CONVOLVE_AVX512)
        s = o.convolve_interp(x, c1, c2, frac, o.CONVOLVE_SCALAR)
        assert abs(a - s) < 1e-5


def test_avx512_stream_equals_the_avx_fma_stream_within_rounding():
    """A whole stream through the AVX-512 leaf: counts identical to the AVX+FMA run (the control flow does not depend on the
    leaf), samples within f32 rounding of it (two 16-lane chains instead of two 8-lane ones)."""
    if not (o.have_avx512f() and o.have_avx_fma()):
        pytest.skip("no AVX-512F on this host")
    from resampler_amd import synth
    x = synth.sweep(40000, 2, 44100.0)
    ya, ca = o.OracleFir(2, 44100, 48000, 128, 90, o.CONVOLVE_AVX512).resample_all(x, 512)
    yb, cb = o.OracleFir(2, 44100, 48000, 128, 90, o.CONVOLVE_AVX_FMA).resample_all(x, 512)
    assert np.array_equal(ca, cb) and ya.size == yb.size
    assert float(np.sqrt(np.mean((ya.astype(np.float64) - yb) ** 2))) < 2e-7


def test_bench_calls_is_the_call_loop():         # benches/benchmark_resampler_fir.rs:50-89
    from resampler_amd import synth
    x = synth.lcg_noise(1024)
    a = o.OracleFir(2, 44100, 48000, 128, 90)
    b = o.OracleFir(2, 44100, 48000, 128, 90)
    out = np.zeros(a.buffer_size_output(), np.float32)
    total = a.bench_calls(x, out, 7)
    want = 0
    for _ in range(7):
        rc, c, p = b.resample(x, out)
        assert rc == 0
        want += p
    assert total == want and a.state() == b.state()


def test_convolve_impulse():                    # fir/mod.rs:194-247
    taps = 32
    x = np.zeros(taps, np.float32)
    x[5] = 1.0
    c1 = np.arange(taps, dtype=np.float32) * np.float32(0.01)
    c2 = np.arange(taps, dtype=np.float32) * np.float32(0.02)
    for kind in (o.CONVOLVE_SCALAR,) + ((o.CONVOLVE_AVX_FMA,) if o.have_avx_fma() else ()):
        got = o.convolve_interp(x, c1, c2, 0.5, kind)
        assert abs(got - (0.05 * 0.5 + 0.10 * 0.5)) < 1e-6


def _impulse_response(in_hz, out_hz, duration=5.0):
    # resampler_fir.rs:693-739: mono impulse at the middle, 256-value chunks.
    n = int(np.float32(in_hz) * np.float32(duration))
    x = np.zeros(n, np.float32)
    x[min(int(n * 0.5), n - 1)] = 1.0
    r = o.OracleFir(1, in_hz, out_hz, 128, 90)
    out, _ = r.resample_all(x, 256)
    return out


@pytest.mark.parametrize("in_hz,out_hz", [(22050, 44100), (22050, 48000)])
def test_stopband_attenuation(in_hz, out_hz, golden):   # resampler_fir.rs:741-815
    y = _impulse_response(in_hz, out_hz)
    peak = int(np.argmax(np.abs(y)))
    win = int(np.float32(out_hz) * np.float32(0.1))
    start = max(0, peak - win // 2)
    seg = y[start:min(start + win, len(y))]
    fft_size = 8192
    spec = np.fft.rfft(seg.astype(np.float64), fft_size) if len(seg) <= fft_size else \
        np.fft.rfft(seg[:fft_size].astype(np.float64), fft_size)
    mag_db = 20.0 * np.log10(np.maximum(np.abs(spec), 1e-10))

    def to_bin(f):
        return int(round(f * fft_size / out_hz))
    nyq = in_hz / 2.0
    pass_max = mag_db[to_bin(20.0):to_bin(nyq * 0.9) + 1].max()
    stop_end = min(len(mag_db) - 10, to_bin(out_hz / 2.0 * 0.95))
    stop_max = mag_db[to_bin(nyq * 1.1):stop_end + 1].max()
    assert pass_max - stop_max >= golden["fir_stopband_min_db"]


def test_constant_input_counts_and_state():     # resampler_fir.rs:817-839 (+ SURVEY 8a4 probe)
    r1 = o.OracleFir(1, 48000, 44100, 128, 90)
    r2 = o.OracleFir(1, 48000, 44100, 128, 90)
    x = np.full(512, 0.5, np.float32)
    o1 = np.zeros(r1.buffer_size_output(), np.float32)
    o2 = np.zeros(r2.buffer_size_output(), np.float32)
    seq = []
    for _ in range(5):
        rc1, c1, p1 = r1.resample(x, o1)
        rc2, c2, p2 = r2.resample(x, o2)
        assert (rc1, c1, p1) == (rc2, c2, p2)
        assert np.array_equal(o1[:p1], o2[:p2])
        seq.append((c1, p1))
    assert seq == [(512, 354), (512, 471), (512, 470), (512, 470), (512, 471)]
    assert r1.buffer_size_output() == 3648
    assert o.OracleFir(2, 44100, 48000, 128, 90).buffer_size_output() == 8642


def test_arbitrary_rates_and_errors():          # resampler_fir.rs:841-862
    r = o.OracleFir(1, 24000, 16000, 64, 60)
    out = np.zeros(r.buffer_size_output(), np.float32)
    assert r.resample(np.zeros(256, np.float32), out)[0] == 0
    with pytest.raises(ValueError):
        o.OracleFir(1, 0, 44100, 128, 120)
    with pytest.raises(ValueError):
        o.OracleFir(1, 44100, 0, 128, 120)
    r3 = o.OracleFir(3, 44100, 48000, 32, 120)
    out3 = np.zeros(r3.buffer_size_output(), np.float32)
    assert r3.resample(np.zeros(100, np.float32), out3)[0] == 1   # InvalidInputBufferSize
    assert r3.resample(np.zeros(99, np.float32), out3[:100])[0] == 2  # InvalidOutputBufferSize


def test_reset_and_compaction_paths():
    # Not covered by the reference's tests (SURVEY section 4 gaps): exercised here so that the
    # oracle and the HIP path are compared on them too.
    r = o.OracleFir(2, 96000, 44100, 128, 120)
    rng = np.random.default_rng(1)
    out = np.zeros(64, np.float32)          # tiny output: input piles up, read_position moves
    total_c = total_p = 0
    for _ in range(400):
        x = rng.standard_normal(2 * 300).astype(np.float32)
        rc, c, p = r.resample(x, out)
        assert rc == 0
        total_c += c
        total_p += p
    rp, av, pos = r.state()
    assert av <= 4096 and rp <= 4096
    r.reset()
    assert r.state() == (0, 0, 0.0)
