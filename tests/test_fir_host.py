"""CPU-side tests of the product's host logic (no GPU, no compute entry points): filter design
tables and the closed-form replay of the ResamplerFir state machine vs the oracle."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

import resampler_amd as ra
from oracle import pyoracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RATE_PAIRS = [(44100, 48000), (48000, 44100), (44100, 96000), (96000, 44100), (48000, 96000),
              (96000, 48000), (22050, 48000), (16000, 44100), (24000, 16000), (44100, 48001),
              (8000, 192000), (384000, 16000), (11025, 192000), (1000003, 999983)]


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "resampler_amd.h")).read()
    declared = set(re.findall(r"\b(rsmp_[a-z0-9_]+)\s*\(", header))
    lib = C.CDLL(ra.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(ra.declared_symbols()), declared ^ set(ra.declared_symbols())


def test_rust_shim_binds_symbols_of_the_header_with_matching_arity():
    """No rustc in the image: the source-only crate (bindings/rust) cannot be compiled here.  What can be checked:
    every `fn rsmp_*` of its extern block names an entry point of the C header, with the same number of arguments,
    and both libraries export it."""
    src = open(os.path.join(ROOT, "bindings", "rust", "src", "lib.rs")).read()
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "resampler_amd.h")).read(), flags=re.S)
    rust = {m.group(1): m.group(2) for m in re.finditer(r"fn (rsmp_[a-z0-9_]+)\s*\(([^)]*)\)", src, flags=re.S)}
    assert len(rust) >= 19
    lib = C.CDLL(ra.LIB_PATH)
    for name, args in rust.items():
        m = re.search(r"\b" + name + r"\s*\(([^)]*)\)", header, flags=re.S)
        assert m, name
        n_rust = len([a for a in args.split(",") if a.strip()])
        c_args = m.group(1).strip()
        n_c = 0 if c_args in ("", "void") else len(c_args.split(","))
        assert n_rust == n_c, (name, args, c_args)
        assert hasattr(lib, name), name


def test_no_device_fails_loudly():
    if ra.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(ra.ResampleError) as e:
        ra.ResamplerFir.new(2, ra.SampleRate.Hz44100, ra.SampleRate.Hz48000)
    assert "no HIP device" in str(e.value)


def test_invalid_arguments():
    L = ra.lib()
    assert not L.rsmp_fir_new_from_hz(2, 0, 44100, 3, 2, 0)
    assert "input sample rate must be greater than zero" in ra.last_error()
    assert not L.rsmp_fir_new_from_hz(2, 44100, 0, 3, 2, 0)
    assert "output sample rate must be greater than zero" in ra.last_error()
    assert not L.rsmp_fir_new(2, 99, 4, 3, 2, 0)
    assert not L.rsmp_fir_new_from_hz(2, 44100, 48000, 7, 2, 0)
    assert [ra.SampleRate(i).hz for i in range(10)] == [22050, 16000, 32000, 44100, 48000, 88200,
                                                       96000, 176400, 192000, 384000]


@pytest.mark.parametrize("in_hz,out_hz", [(44100, 48000), (48000, 44100), (96000, 44100), (24000, 16000)])
@pytest.mark.parametrize("latency,att", [(ra.Latency.Sample64, ra.Attenuation.Db90),
                                         (ra.Latency.Sample16, ra.Attenuation.Db60),
                                         (ra.Latency.Sample8, ra.Attenuation.Db120)])
def test_design_table_is_bit_identical_to_oracle(in_hz, out_hz, latency, att):
    mine = ra.design_fir_coeffs(in_hz, out_hz, latency, att)
    ref = o.OracleFir(1, in_hz, out_hz, latency.taps(), (60, 90, 120)[int(att)]).coeffs()
    assert np.array_equal(mine.view(np.uint32), ref.view(np.uint32))


def test_cutoff(golden):
    for n, expected in golden["cutoff_kaiser_beta10"]:
        assert abs(ra.design_cutoff_kaiser(n, 10.0) / expected - 1) < 1e-6


def _naive_positions(position, ratio, taps, available, cap):
    """The reference loop (resampler_fir.rs:542-590) as a plain Python f64 recurrence."""
    ps = []
    while True:
        off = math.floor(position)
        if off + taps > available or len(ps) >= cap:
            break
        ps.append(position)
        position += ratio
    return ps, position


@pytest.mark.parametrize("in_hz,out_hz", RATE_PAIRS)
def test_plan_matches_oracle_counts_and_state(in_hz, out_hz):
    rng = np.random.default_rng(in_hz * 7 + out_hz)
    for taps, lat in ((128, ra.Latency.Sample64), (16, ra.Latency.Sample8)):
        ref = o.OracleFir(1, in_hz, out_hz, taps, 90)
        plan = ra.FirPlan(in_hz, out_hz, lat)
        out = np.zeros(ref.buffer_size_output(), np.float32)
        for step in range(60):
            n_in = int(rng.choice([0, 1, 17, 256, 512, 1000, 4096, 5000]))
            cap = int(rng.choice([out.size, out.size, 64, 0, 300]))
            rc, c, p = ref.resample(np.zeros(n_in, np.float32), out[:cap])
            a, pr = plan.call(n_in, cap)
            assert rc == 0 and (a, pr) == (c, p), (step, n_in, cap)
            assert plan.state() == ref.state()
        ref.reset()
        plan.reset()
        assert plan.state() == ref.state() == (0, 0, 0.0)


@pytest.mark.parametrize("in_hz,out_hz", RATE_PAIRS)
def test_plan_segments_reproduce_the_f64_recurrence_exactly(in_hz, out_hz):
    taps = 128
    plan = ra.FirPlan(in_hz, out_hz, ra.Latency.Sample64)
    ratio = in_hz / out_hz
    rng = np.random.default_rng(3)
    read_pos, avail, position = 0, 0, 0.0
    for step in range(25):
        n_in = int(rng.choice([64, 256, 700, 4096]))
        cap = int(rng.choice([100000, 100000, 90]))
        # python mirror of the bookkeeping (resampler_fir.rs:524-528, 596-615)
        acc = min(n_in, max(0, 8192 - (read_pos + avail)), 4096 - avail)
        avail += acc
        expect, end_pos = _naive_positions(position, ratio, taps, avail, cap)
        a, p, segs = plan.call(n_in, cap, want_segments=True)
        assert (a, p) == (acc, len(expect))
        got = []
        nxt = 0
        for out_start, count, in_base, p0, inc in segs:
            assert out_start == nxt and count >= 1
            nxt += count
            k = np.arange(count, dtype=np.float64)
            got.extend((p0 + k * inc).tolist())
        assert got == expect          # bit-exact f64 equality, every output frame
        consumed = min(math.floor(end_pos), avail)
        read_pos += consumed
        avail -= consumed
        position = end_pos - consumed
        if read_pos > 4096:
            read_pos = 0
        assert plan.state() == (read_pos, avail, position)
    # a call is a handful of runs, not hundreds of adds
    plan2 = ra.FirPlan(in_hz, out_hz, ra.Latency.Sample64)
    plan2.call(4096, 10 ** 6)
    a, p, segs = plan2.call(4096, 10 ** 6, want_segments=True)
    if p > 200 and ratio < 64:
        assert len(segs) < 60, len(segs)


def test_run_planner_fast_path_equals_the_plain_state_machine():
    """The device planner of rsmp_fir_lockstep_run (fir_mirror_fast.h: call structure predicted in exact integer arithmetic,
    the f64 chain checked against it, outputs at integer positions by replay) against mirror_call on the host, call by
    call: counts, every bit of the state, the outputs taking the row-1023 variant -- all ordered pairs of twelve rates x
    five (latency, call size, run length) shapes, ~1 M calls, from a non-fresh state."""
    import itertools
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    from selftest_fast_planner import RATES, SHAPES, selftest
    total = slow_total = 0
    for i, o_ in itertools.permutations(RATES, 2):
        for lat, frames, calls, runlen in SHAPES:
            bad, slow, lean = selftest(i, o_, lat, frames, calls, runlen, prefeed=(i // 100) % 300)
            assert bad == 0, (i, o_, lat, frames, bad)
            total += calls
            slow_total += slow
    # the fast path must be the rule (what it declines: calls of > 65535 outputs inside one binade)
    assert slow_total * 100 < total, (slow_total, total)
    # BASELINE config 4's pairs at its call size: never declined
    for i, o_ in [(44100, 48000), (48000, 44100), (44100, 96000), (96000, 44100), (48000, 96000), (96000, 48000)]:
        bad, slow, lean = selftest(i, o_, 3, 512, 20000, 256)
        assert (bad, slow) == (0, 0) and lean > 15000, (i, o_, bad, slow, lean)   # (and mostly by the unchecked chain)
