"""GPU parity tests of rsmp_fir_lockstep_run: k consecutive resample() calls per stream (src/resampler_fir.rs:509-621,
driven like resample/src/main.rs:226-254) planned on the device and computed by the bulk kernels in one go -- the same
calls, (consumed, produced) per call, samples (1e-6 RMS against the CPU oracle's AVX+FMA path) and end state (bit for
bit) as k lock-step steps."""
import os

import numpy as np
import pytest

import resampler_amd as ra
from oracle import pyoracle as o
from resampler_amd import sharding, synth

pytestmark = pytest.mark.gpu

RMS_TOL = 1e-6   # north_star tolerance
ATT_DB = {ra.Attenuation.Db60: 60, ra.Attenuation.Db90: 90, ra.Attenuation.Db120: 120}


def rms(a, b):
    if a.size == 0:
        return 0.0
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


def drive(specs, schedule, frames, seed=0, lat=ra.Latency.Sample64, att=ra.Attenuation.Db90, prefeed=None, xs=None,
          relative=False, check_state=True):
    """`schedule`: a list of ("run", k) / ("step",) / ("reset",) entries executed in order on one lock-step batch over
    `specs`; every call of every stream is mirrored by an OracleFir.  Asserts identical counts for every call, returns
    the worst RMS error over the streams (relative to each stream's level with `relative`)."""
    import torch
    dev = torch.device("cuda:0")
    n = len(specs)
    hs = [ra.ResamplerFir.new_from_hz(s.channels, s.in_hz, s.out_hz, lat, att) for s in specs]
    kind = o.CONVOLVE_AVX_FMA if o.have_avx_fma() else o.CONVOLVE_SCALAR
    refs = [o.OracleFir(s.channels, s.in_hz, s.out_hz, lat.taps(), ATT_DB[att], kind) for s in specs]
    rng = np.random.default_rng(seed)
    if prefeed is not None:
        for i, (h, r) in enumerate(zip(hs, refs)):
            if prefeed[i] == 0:
                continue
            x = (rng.random(prefeed[i] * specs[i].channels, dtype=np.float32) * 2 - 1).astype(np.float32)
            og = np.zeros(h.buffer_size_output(), np.float32)
            orr = np.zeros(r.buffer_size_output(), np.float32)
            off = 0
            while off < x.size:
                cg, pg = h.resample(x[off:off + 300 * specs[i].channels], og)
                rc, cr, pr = r.resample(x[off:off + 300 * specs[i].channels], orr)
                assert rc == 0 and (cg, pg) == (cr, pr)
                off += cg
    total_calls = sum(e[1] if e[0] == "run" else (1 if e[0] == "step" else 0) for e in schedule)
    if xs is None:
        xs = [(rng.random(total_calls * frames * s.channels, dtype=np.float32) * 2 - 1).astype(np.float32) for s in specs]
    d_in = [torch.from_numpy(x).to(dev) for x in xs]
    caps = [h.buffer_size_output() for h in hs]
    d_out = [torch.zeros(total_calls * c, device=dev, dtype=torch.float32) for c in caps]
    ls = ra.FirLockstep(hs, frames)
    ls.bind_caps(d_in, d_out, caps)
    ref_out = [[] for _ in range(n)]
    orr = [np.zeros(r.buffer_size_output(), np.float32) for r in refs]
    call = 0

    def oracle_calls(k, cons, prod):
        nonlocal call
        for s in range(k):
            for i, sp in enumerate(specs):
                sl = xs[i][(call + s) * frames * sp.channels:(call + s + 1) * frames * sp.channels]
                rc, cr, pr = refs[i].resample(sl, orr[i])
                assert rc == 0
                assert (int(cons[s][i]), int(prod[s][i])) == (cr, pr), (call + s, i, (cons[s][i], prod[s][i]), (cr, pr))
                ref_out[i].append(orr[i][:pr].copy())
        call += k

    worst = 0.0

    def compare():
        nonlocal worst
        for i in range(n):
            want = np.concatenate(ref_out[i]) if ref_out[i] else np.zeros(0, np.float32)
            got = d_out[i][:want.size].cpu().numpy()
            err = rms(got, want)
            if relative:
                err /= max(float(np.sqrt(np.mean(want.astype(np.float64) ** 2))), 1e-300)
            if err > RMS_TOL:
                bad = np.flatnonzero(~(np.abs(got.astype(np.float64) - want) <= 1e-4))
                print(f"run: stream {i} {specs[i]} rms {err:.3e} bad {bad.size}: {bad[:8]} got {got[bad[:4]]} want {want[bad[:4]]}")
            worst = max(worst, err)

    for e in schedule:
        if e[0] == "run":
            ls.run(e[1], frames, call * frames, append=True)
            cons, prod = ls.run_counts()
            assert cons.shape == (e[1], n)
            lc, lp = ls.counts()   # the last call's counts, as after a step
            assert np.array_equal(lc, cons[-1]) and np.array_equal(lp, prod[-1])
            oracle_calls(e[1], cons, prod)
        elif e[0] == "step":
            ls.step(frames, call * frames, append=True)
            cons, prod = ls.counts()
            oracle_calls(1, cons[None, :], prod[None, :])
        elif e[0] == "reset":   # (also takes the append position back to the front of the output buffers)
            compare()
            ls.reset()
            for i, r in enumerate(refs):
                r.reset()
                ref_out[i] = []
    compare()
    if check_state:
        ls.sync()
        for h, r in zip(hs, refs):
            assert h.state() == r.state()
    return worst, ls, hs, refs


@pytest.mark.parametrize("k", [1, 2, 16, 40])
def test_run_of_k_calls_equals_k_steps(k):
    specs = sharding.mixed_rate_batch(42, 2, 512)
    worst, ls, hs, refs = drive(specs, [("run", k), ("run", k)], 512, seed=k)
    assert worst <= RMS_TOL, worst
    assert not ls.status().any()
    ls.close()


def test_runs_and_steps_interleave_on_carried_state():
    """step, run, step, step, run, reset, run: the history buffers alternate per launch whatever its kind, plans the
    one-call kernel made ahead are dropped after a run, the append position carries across."""
    specs = sharding.mixed_rate_batch(24, 2, 512)
    sched = [("step",), ("run", 5), ("step",), ("step",), ("run", 3), ("run", 1), ("step",), ("reset",), ("run", 4), ("step",)]
    worst, ls, hs, refs = drive(specs, sched, 512, seed=3)
    assert worst <= RMS_TOL, worst
    ls.close()


def test_streams_in_different_states_and_short_calls():
    specs = sharding.mixed_rate_batch(36, 2, 200)
    prefeed = [(37 * i) % 900 for i in range(36)]
    worst, ls, hs, refs = drive(specs, [("run", 7), ("run", 9)], 200, seed=5, prefeed=prefeed)
    assert worst <= RMS_TOL, worst
    ls.close()


@pytest.mark.parametrize("channels", [1, 3, 4, 8])
def test_other_channel_counts(channels):
    specs = sharding.mixed_rate_batch(12, channels, 512)
    worst, ls, hs, refs = drive(specs, [("run", 6), ("step",), ("run", 6)], 512, seed=channels)
    assert worst <= RMS_TOL, worst
    ls.close()


def test_other_tap_counts_and_attenuations():
    specs = [sharding.StreamSpec(2, i, o_) for i, o_ in [(44100, 48000), (48000, 44100), (32000, 48000), (48000, 16000), (22050, 44100)]] * 3
    for lat, att in [(ra.Latency.Sample16, ra.Attenuation.Db60), (ra.Latency.Sample32, ra.Attenuation.Db120), (ra.Latency.Sample8, ra.Attenuation.Db90)]:
        worst, ls, hs, refs = drive(specs, [("run", 9), ("run", 4)], 384, seed=11, lat=lat, att=att)
        assert worst <= RMS_TOL, (lat, att, worst)
        ls.close()


def test_irrational_ratios_fall_back_to_a_loop_of_steps():
    specs = [sharding.StreamSpec(2, 44100, 48000), sharding.StreamSpec(2, 44101, 47999), sharding.StreamSpec(2, 48000, 44100)]
    worst, ls, hs, refs = drive(specs, [("run", 5), ("run", 3)], 512, seed=2)
    assert worst <= RMS_TOL, worst
    ls.close()


def test_levels_from_1e_minus_30_to_3e4_keep_their_relative_precision():
    specs = sharding.mixed_rate_batch(42, 2, 512)
    levels = [1.0, 2.0 ** -10, 2.0 ** -17, 2.0 ** -24, 1e-30, 3e4, 1.0]
    rng = np.random.default_rng(123)
    k = 12
    xs = []
    for i, sp in enumerate(specs):
        x = (rng.random(2 * k * 512 * 2, dtype=np.float32) * 2 - 1).astype(np.float32) * np.float32(levels[(i // 6) % 7])
        if (i // 6) % 7 == 6:
            x[k * 512:] *= np.float32(2.0 ** -20)
        xs.append(x.astype(np.float32))
    worst, ls, hs, refs = drive(specs, [("run", k), ("run", k)], 512, xs=xs, relative=True)
    assert worst <= RMS_TOL, worst
    ls.close()


def test_non_finite_samples_match_the_reference():
    """inf / NaN / huge samples inside a run: the same finite / inf / NaN pattern as the reference (the bulk kernels mark
    the chunk, the repair launch redoes it in the reference's two-row form)."""
    import torch
    dev = torch.device("cuda:0")
    specs = sharding.mixed_rate_batch(12, 2, 512)
    k = 8
    rng = np.random.default_rng(7)
    xs = []
    for i, sp in enumerate(specs):
        x = (rng.random(k * 512 * 2, dtype=np.float32) * 2 - 1).astype(np.float32)
        x[2 * (700 + 13 * i)] = [np.inf, -np.inf, np.nan, 3e38, 1e30, 70000.0][i % 6]
        xs.append(x)
    hs = [ra.ResamplerFir.new_from_hz(2, s.in_hz, s.out_hz, ra.Latency.Sample64, ra.Attenuation.Db90) for s in specs]
    kind = o.CONVOLVE_AVX_FMA if o.have_avx_fma() else o.CONVOLVE_SCALAR
    refs = [o.OracleFir(2, s.in_hz, s.out_hz, 128, 90, kind) for s in specs]
    caps = [h.buffer_size_output() for h in hs]
    d_in = [torch.from_numpy(x).to(dev) for x in xs]
    d_out = [torch.zeros(k * c, device=dev) for c in caps]
    ls = ra.FirLockstep(hs, 512)
    ls.bind_caps(d_in, d_out, caps)
    ls.run(k, 512, 0)
    cons, prod = ls.run_counts()
    for i, r in enumerate(refs):
        out = np.zeros(caps[i], np.float32)
        ys = []
        for s in range(k):
            rc, c, p = r.resample(xs[i][s * 1024:(s + 1) * 1024], out)
            assert rc == 0 and (c, p) == (int(cons[s][i]), int(prod[s][i]))
            ys.append(out[:p].copy())
        want = np.concatenate(ys)
        got = d_out[i][:want.size].cpu().numpy()
        fin = np.isfinite(want)
        # the documented residue: inf versus NaN where the reference's answer hangs on a ~1e-12 rounding residue
        assert np.array_equal(np.isfinite(got), fin), i
        assert rms(got[fin], want[fin]) <= RMS_TOL * max(1.0, float(np.max(np.abs(want[fin])))), i
    ls.close()


def test_without_append_a_run_starts_at_the_front_of_the_output():
    import torch
    dev = torch.device("cuda:0")
    specs = sharding.mixed_rate_batch(6, 2, 512)
    hs = [ra.ResamplerFir.new_from_hz(2, s.in_hz, s.out_hz, ra.Latency.Sample64, ra.Attenuation.Db90) for s in specs]
    refs = [o.OracleFir(2, s.in_hz, s.out_hz, 128, 90) for s in specs]
    xs = [synth.hash_noise(8 * 512 * 2, seed=i) for i in range(6)]
    caps = [h.buffer_size_output() for h in hs]
    d_in = [torch.from_numpy(x).to(dev) for x in xs]
    d_out = [torch.zeros(4 * c, device=dev) for c in caps]
    ls = ra.FirLockstep(hs, 512)
    ls.bind_caps(d_in, d_out, caps)
    for half in range(2):
        ls.run(4, 512, half * 4 * 512, append=False)
        cons, prod = ls.run_counts()
        for i, r in enumerate(refs):
            out = np.zeros(caps[i], np.float32)
            ys = []
            for s in range(4):
                rc, c, p = r.resample(xs[i][(half * 4 + s) * 1024:(half * 4 + s + 1) * 1024], out)
                assert (c, p) == (int(cons[s][i]), int(prod[s][i]))
                ys.append(out[:p].copy())
            want = np.concatenate(ys)
            assert rms(d_out[i][:want.size].cpu().numpy(), want) <= RMS_TOL
    ls.close()


def test_runs_in_a_row_are_planned_ahead_and_still_exact():
    """From the third run of one shape on, the next run is planned on a stream of its own while the current one computes
    (guessed to repeat the shape, its input offset moving on as before) and taken over when it is really asked for:
    same counts, samples and states; a guess that does not come true (a step in between, another k, a reset) is dropped."""
    specs = sharding.mixed_rate_batch(30, 2, 512)
    sched = [("run", 5)] * 6 + [("step",)] + [("run", 5)] * 4 + [("run", 3)] * 4 + [("reset",)] + [("run", 6)] * 5 + [("step",), ("step",)]
    worst, ls, hs, refs = drive(specs, sched, 512, seed=21)
    assert worst <= RMS_TOL, worst
    assert not ls.status().any()
    ls.close()


def test_the_same_span_run_again_and_again_without_append():
    """The bench's pattern: every run reads the SAME resident span (offset 0) on carried state and starts at the front of
    the output -- the run planned ahead guesses an unchanged offset from the second repetition on."""
    import torch
    dev = torch.device("cuda:0")
    specs = sharding.mixed_rate_batch(18, 2, 512)
    k = 8
    hs = [ra.ResamplerFir.new_from_hz(2, s.in_hz, s.out_hz, ra.Latency.Sample64, ra.Attenuation.Db90) for s in specs]
    kind = o.CONVOLVE_AVX_FMA if o.have_avx_fma() else o.CONVOLVE_SCALAR
    refs = [o.OracleFir(2, s.in_hz, s.out_hz, 128, 90, kind) for s in specs]
    xs = [synth.hash_noise(k * 512 * 2, seed=40 + i) for i in range(len(specs))]
    caps = [h.buffer_size_output() for h in hs]
    d_in = [torch.from_numpy(x).to(dev) for x in xs]
    d_out = [torch.zeros(k * c, device=dev) for c in caps]
    ls = ra.FirLockstep(hs, 512)
    ls.bind_caps(d_in, d_out, caps)
    for rep in range(7):
        ls.run(k, 512, 0, append=False)
        cons, prod = ls.run_counts()
        for i, r in enumerate(refs):
            out = np.zeros(caps[i], np.float32)
            ys = []
            for s in range(k):
                rc, c, p = r.resample(xs[i][s * 1024:(s + 1) * 1024], out)
                assert rc == 0 and (c, p) == (int(cons[s][i]), int(prod[s][i])), (rep, i, s)
                ys.append(out[:p].copy())
            want = np.concatenate(ys)
            assert rms(d_out[i][:want.size].cpu().numpy(), want) <= RMS_TOL, (rep, i)
    ls.sync()
    for h, r in zip(hs, refs):
        assert h.state() == r.state()
    ls.close()


@pytest.mark.parametrize("own_stream", [True, False])
def test_a_big_batch_runs_in_a_row(own_stream):
    """Config 4's path proper: from 256 streams on, K1 of the next run goes in FRONT of a run's bulk kernels on the caller's
    stream and commits the run planned ahead on its way, the caller's stream waits for the planner behind the split launch,
    the item tables come from the plan stream, and -- on a stream of the caller's own -- the launches complete the events
    themselves (fir_lockstep_api.cpp, round 6).  264 streams, runs of 8 calls over the same span again and again, a step in
    between (the plan made ahead is dropped, its commit still due): counts, samples, states."""
    import torch
    dev = torch.device("cuda:0")
    n, k = 264, 8
    specs = sharding.mixed_rate_batch(n, 2, 512)
    hs = [ra.ResamplerFir.new_from_hz(2, s.in_hz, s.out_hz, ra.Latency.Sample64, ra.Attenuation.Db90) for s in specs]
    kind = o.CONVOLVE_AVX_FMA if o.have_avx_fma() else o.CONVOLVE_SCALAR
    refs = [o.OracleFir(2, s.in_hz, s.out_hz, 128, 90, kind) for s in specs]
    xs = [synth.hash_noise(k * 512 * 2, seed=300 + i) for i in range(n)]
    caps = [h.buffer_size_output() for h in hs]
    d_in = [torch.from_numpy(x).to(dev) for x in xs]
    d_out = [torch.zeros(k * c, device=dev) for c in caps]
    ls = ra.FirLockstep(hs, 512)
    ls.bind_caps(d_in, d_out, caps)
    st = torch.cuda.Stream() if own_stream else None
    handle = st.cuda_stream if own_stream else ra.STREAM_LEGACY
    out = [np.zeros(c, np.float32) for c in caps]

    def check_run(rep):
        if st:
            st.synchronize()
        cons, prod = ls.run_counts()
        for i, r in enumerate(refs):
            ys = []
            for c in range(k):
                rc, cc, pp = r.resample(xs[i][c * 1024:(c + 1) * 1024], out[i])
                assert rc == 0 and (cc, pp) == (int(cons[c][i]), int(prod[c][i])), (rep, i, c)
                ys.append(out[i][:pp].copy())
            if i % 7 == rep % 7:   # (the samples of every seventh stream, another seven each time)
                want = np.concatenate(ys)
                assert rms(d_out[i][:want.size].cpu().numpy(), want) <= RMS_TOL, (rep, i)

    for rep in range(5):
        ls.run(k, 512, 0, append=False, stream=handle)
        check_run(rep)
    ls.step(512, 0, append=False, stream=handle)   # (a plan was made ahead for a sixth run: dropped)
    if st:
        st.synchronize()
    c1, p1 = ls.counts()
    for i, r in enumerate(refs):
        rc, cc, pp = r.resample(xs[i][:1024], out[i])
        assert rc == 0 and (cc, pp) == (int(c1[i]), int(p1[i])), i
    for rep in range(5, 9):
        ls.run(k, 512, 0, append=False, stream=handle)
        check_run(rep)
    stats = ls.stats()
    if os.environ.get("RSMP_LS_AHEAD") != "0":   # (tests/test_knobs_gpu.py runs this without the plan stream as well)
        assert stats["plan_ahead_hits"] >= 3, stats   # (the first runs of a shape probe for a plan stream and have nothing to repeat)
    ls.sync()
    for h, r in zip(hs, refs):
        assert h.state() == r.state()
    ls.close()


def test_runs_ordered_on_the_legacy_default_stream():
    """What a torch caller under its default stream passes (ra.torch_stream() == RSMP_STREAM_LEGACY): from the third
    run on the caller's stream WAITS for the plan stream's event -- this image's hipStreamWaitEvent dereferences the
    legacy handle (a segfault found by tools/run_bulk_probe.py), so the library hands it the null stream instead
    (common.h stream_wait_event)."""
    import torch
    dev = torch.device("cuda:0")
    assert ra.torch_stream() == ra.STREAM_LEGACY
    specs = sharding.mixed_rate_batch(12, 2, 512)
    k = 6
    hs = [ra.ResamplerFir.new_from_hz(2, s.in_hz, s.out_hz, ra.Latency.Sample64, ra.Attenuation.Db90) for s in specs]
    refs = [o.OracleFir(2, s.in_hz, s.out_hz, 128, 90) for s in specs]
    xs = [synth.hash_noise(k * 512 * 2, seed=70 + i) for i in range(len(specs))]
    caps = [h.buffer_size_output() for h in hs]
    d_in = [torch.from_numpy(x).to(dev) for x in xs]
    d_out = [torch.zeros(k * c, device=dev) for c in caps]
    ls = ra.FirLockstep(hs, 512)
    ls.bind_caps(d_in, d_out, caps)
    for rep in range(5):
        ls.run(k, 512, 0, append=False, stream=ra.STREAM_LEGACY)
        got = [t.cpu().numpy() for t in d_out]   # (torch's copy on the default stream: ordered behind the run)
        cons, prod = ls.run_counts()
        for i, r in enumerate(refs):
            out = np.zeros(caps[i], np.float32)
            ys = []
            for s in range(k):
                rc, c, p = r.resample(xs[i][s * 1024:(s + 1) * 1024], out)
                assert rc == 0 and (c, p) == (int(cons[s][i]), int(prod[s][i])), (rep, i, s)
                ys.append(out[:p].copy())
            want = np.concatenate(ys)
            assert rms(got[i][:want.size], want) <= RMS_TOL, (rep, i)
    ls.close()


def test_an_old_stream_keeps_parity_through_runs():
    """The reference's f64 position drifts away from the exact rational one by ~1e-14 of a frame per output
    (src/resampler_fir.rs:589: every add rounds on the grid of its binade); after half an hour of audio that is 1e-6 of a
    frame -- 2e-6 of a full-scale sample with coefficient rows mixed for the position without the drift.  A stream that old
    (aged through the bulk path, which follows the drift) must come through runs and steps like a young one."""
    import torch
    dev = torch.device("cuda:0")
    g = ra.ResamplerFir.new_from_hz(2, 44100, 48000, ra.Latency.Sample64, ra.Attenuation.Db90)
    kind = o.CONVOLVE_AVX_FMA if o.have_avx_fma() else o.CONVOLVE_SCALAR
    r = o.OracleFir(2, 44100, 48000, 128, 90, kind)
    x = synth.fast_noise(2 * (4 << 20), seed=21)
    for _ in range(32):   # 134 M frames: 51 minutes at 44.1 kHz
        _, consumed = g.resample_bulk(x, 1024)
        r.resample_all(x, 1024)
        assert consumed == x.size
    assert g.state() == r.state()
    frames, k = 512, 48
    xs = synth.fast_noise(2 * frames * (2 * k + 1), seed=22)
    d_in = [torch.from_numpy(xs).to(dev)]
    cap = g.buffer_size_output()
    d_out = [torch.zeros((2 * k + 1) * cap, device=dev)]
    ls = ra.FirLockstep([g], frames)
    ls.bind_caps(d_in, d_out, [cap])
    want = []
    orr = np.zeros(r.buffer_size_output(), np.float32)

    def oracle(calls, first, cons, prod):
        for s in range(calls):
            rc, cr, pr = r.resample(xs[(first + s) * frames * 2:(first + s + 1) * frames * 2], orr)
            assert rc == 0 and (int(cons[s][0]), int(prod[s][0])) == (cr, pr)
            want.append(orr[:pr].copy())
    assert ls.table_rebinds() == 0   # (created on the old stream: its tables are mixed for its drift from the start)
    ls.run(k, frames, 0, append=True)
    oracle(k, 0, *ls.run_counts())
    ls.step(frames, k * frames, append=True)
    c, p = ls.counts()
    oracle(1, k, c[None, :], p[None, :])
    ls.run(k, frames, (k + 1) * frames, append=True)
    oracle(k, k + 1, *ls.run_counts())
    w = np.concatenate(want)
    got = d_out[0][:w.size].cpu().numpy()
    assert rms(got, w) <= RMS_TOL, rms(got, w)
    ls.sync()
    assert g.state() == r.state()


def test_class_tables_follow_the_drift_while_a_batch_runs():
    """Two streams run 17 M frames in lock-step (runs of 256 calls, a few steps in between): the drift moves by more than
    the tolerance of a set of class tables, the batch replaces them on the way (table_rebinds), counts, samples and states
    keep matching the oracle; a reset takes the streams -- and the tables -- back to drift 0."""
    import torch
    dev = torch.device("cuda:0")
    specs = [(44100, 48000), (48000, 44100)]
    hs = [ra.ResamplerFir.new_from_hz(2, i, o_, ra.Latency.Sample64, ra.Attenuation.Db90) for i, o_ in specs]
    kind = o.CONVOLVE_AVX_FMA if o.have_avx_fma() else o.CONVOLVE_SCALAR
    refs = [o.OracleFir(2, i, o_, 128, 90, kind) for i, o_ in specs]
    frames, k, runs = 512, 256, 128
    x = synth.fast_noise(2 * frames * (k + 1), seed=31)     # the same span of input again and again
    d_in = [torch.from_numpy(x).to(dev) for _ in hs]
    caps = [h.buffer_size_output() for h in hs]
    d_out = [torch.zeros((k + 1) * c, device=dev) for c in caps]
    ls = ra.FirLockstep(hs, frames)
    ls.bind_caps(d_in, d_out, caps)
    orr = [np.zeros(r.buffer_size_output(), np.float32) for r in refs]

    def one_pass(check):
        ls.run(k, frames, 0, append=False)
        cons, prod = ls.run_counts()
        ls.step(frames, k * frames, append=True)
        c1, p1 = ls.counts()
        worst = 0.0
        for i, r in enumerate(refs):
            want = []
            for s in range(k + 1):
                rc, cr, pr = r.resample(x[s * frames * 2:(s + 1) * frames * 2], orr[i])
                cg, pg = (cons[s][i], prod[s][i]) if s < k else (c1[i], p1[i])
                assert rc == 0 and (int(cg), int(pg)) == (cr, pr), (s, i)
                if check:
                    want.append(orr[i][:pr].copy())
            if check:
                w = np.concatenate(want)
                worst = max(worst, rms(d_out[i][:w.size].cpu().numpy(), w))
        return worst
    for p in range(runs):
        worst = one_pass(check=(p % 16 == 15))
        assert worst <= RMS_TOL, (p, worst)
    assert ls.table_rebinds() >= 2   # (both rate pairs have moved by more than 1.2e-7 of a frame)
    ls.sync()
    for h, r in zip(hs, refs):
        assert h.state() == r.state()
    before = ls.table_rebinds()
    ls.reset()
    for r in refs:
        r.reset()
    assert ls.table_rebinds() > before   # back to the tables of drift 0
    assert one_pass(check=True) <= RMS_TOL


def test_table_replacements_stay_off_the_launch_path():
    """a class of streams that crosses its drift tolerance gets new class tables WITHOUT the
    launch call building, allocating, copying or waiting (fir_table_refresher.h: a worker thread fills the other of
    two owned device images, the crossing swaps pointers with one patch kernel).  Six rate pairs run 96 runs of 256
    calls with the tolerance at its minimum (2e-8 of a frame, looked at after every run), so every class is replaced
    several times: every run call -- also those that swap tables -- returns in well under a tenth of a millisecond
    of host time, nobody ever waits for the worker, the calls' counts and the samples keep matching the oracle."""
    import time
    import torch
    dev = torch.device("cuda:0")
    pairs = [(44100, 48000), (48000, 44100), (44100, 96000), (96000, 44100), (48000, 96000), (96000, 48000)]
    hs = [ra.ResamplerFir.new_from_hz(2, i, o_, ra.Latency.Sample64, ra.Attenuation.Db90) for i, o_ in pairs]
    kind = o.CONVOLVE_AVX_FMA if o.have_avx_fma() else o.CONVOLVE_SCALAR
    refs = [o.OracleFir(2, i, o_, 128, 90, kind) for i, o_ in pairs]
    frames, k, runs = 512, 256, 96
    x = synth.fast_noise(2 * frames * k, seed=77)          # the same span of input again and again
    d_in = [torch.from_numpy(x).to(dev) for _ in hs]
    caps = [h.buffer_size_output() for h in hs]
    d_out = [torch.zeros(k * c, device=dev) for c in caps]
    ls = ra.FirLockstep(hs, frames)
    ls.bind_caps(d_in, d_out, caps)
    ls.set_drift_policy(2e-8, 1)
    stream = torch.cuda.Stream()
    host_us, swapped = [], []
    worst = 0.0
    with torch.cuda.stream(stream):
        for p in range(runs):
            before = ls.stats()["table_rebinds"]
            t0 = time.perf_counter()
            ls.run(k, frames, 0, append=False, stream=stream.cuda_stream)
            host_us.append((time.perf_counter() - t0) * 1e6)
            swapped.append(ls.stats()["table_rebinds"] > before)
            stream.synchronize()
            check = p % 12 == 11 or swapped[-1]
            cons, prod = ls.run_counts() if check else (None, None)
            for i, r in enumerate(refs):
                y, calls = r.resample_all(x, 2 * frames, max_calls=k + 4)
                if check:
                    assert calls.shape[0] == k and (calls[:, 0] == cons[:, i]).all() and (calls[:, 1] == prod[:, i]).all(), (p, i)
                    worst = max(worst, rms(d_out[i][:y.size].cpu().numpy(), y))
    st = ls.stats()
    assert worst <= RMS_TOL, worst
    assert st["table_rebinds"] >= 12 and sum(swapped) >= 6, st      # (every class replaced, most of them more than once)
    assert st["table_waits"] == 0, st
    # (the plan stream is found by the device-side probe on every box seen so far; a box whose queues leave none beside
    # the caller's stream runs without plan-ahead, which is slower, not wrong)
    assert st["plan_ahead_hits"] >= runs - 12 or not st["has_plan_stream"], st
    warm = np.array(host_us[8:])
    swap_calls = np.array([h for h, s in zip(host_us[8:], swapped[8:]) if s])
    # Measured: 40-50 us for a run call, 70-90 us for one that swaps tables (seven launches + events through ctypes).  The
    # bounds leave room for a slow or busy host; what they exclude is the launch path this replaces, which held the thread
    # for 3-6 ms per crossing (0.8 ms per run on the driver's box of round 4).
    assert np.median(warm) < 250.0, (np.median(warm), warm.max())
    assert swap_calls.size and np.median(swap_calls) < 300.0 and swap_calls.max() < 2000.0, swap_calls
    ls.sync()
    for h, r in zip(hs, refs):
        assert h.state() == r.state()


@pytest.mark.parametrize("calls,chunk,fresh", [(1100, 64, False), (4096, 32, False), (2000, 48, True), (1024, 128, False)])
def test_long_runs_of_few_streams(calls, chunk, fresh):
    """Bulk launches of thousands of calls per stream (rsmp_fir_lockstep_run_bulk: 16 .. 64 chunks of 64 calls for the
    planner's chain, whose parallel path takes a chunk at a time -- a run of lean calls -- and leaves the call in ~160
    with an output exactly at its limit to the checked path).  Twelve streams of six rate pairs in twelve states (`fresh`:
    straight from reset), launch after launch on the streams' own state: every call's counts, the samples, the end states."""
    import torch
    dev = torch.device("cuda:0")
    n = 12
    specs = sharding.mixed_rate_batch(n, 2, 512)
    hs = [ra.ResamplerFir.new_from_hz(2, s.in_hz, s.out_hz, ra.Latency.Sample64, ra.Attenuation.Db90) for s in specs]
    kind = o.CONVOLVE_AVX_FMA if o.have_avx_fma() else o.CONVOLVE_SCALAR
    refs = [o.OracleFir(2, s.in_hz, s.out_hz, 128, 90, kind) for s in specs]
    rng = np.random.default_rng(calls)
    if not fresh:
        for i, (h, r) in enumerate(zip(hs, refs)):   # distinct states
            x = (rng.random(2 * (300 + 41 * i), dtype=np.float32) * 2 - 1).astype(np.float32)
            og, orr = np.zeros(h.buffer_size_output(), np.float32), np.zeros(r.buffer_size_output(), np.float32)
            off = 0
            while off < x.size:
                cg, pg = h.resample(x[off:off + 2 * 173], og)
                rc, cr, pr = r.resample(x[off:off + 2 * 173], orr)
                assert rc == 0 and (cg, pg) == (cr, pr)
                off += cg
    total = calls * chunk
    caps = [h.buffer_size_output() for h in hs]
    ls = ra.FirLockstep(hs, 512)
    for launch in range(3):
        xs = [(rng.random(2 * total, dtype=np.float32) * 2 - 1).astype(np.float32) for _ in range(n)]
        d_in = [torch.from_numpy(x).to(dev) for x in xs]
        d_out = [torch.zeros((calls + 2) * c, device=dev) for c in caps]
        ls.bind_caps(d_in, d_out, caps)
        ls.run_bulk(total, chunk)
        cons, prod = ls.run_counts()
        worst = 0.0
        for i, r in enumerate(refs):
            y, cl = r.resample_all(xs[i], 2 * chunk, max_calls=calls + 4)
            assert cl.shape[0] == calls, (i, cl.shape)
            assert (cl[:, 0] == cons[:, i]).all() and (cl[:, 1] == prod[:, i]).all(), (launch, i)
            worst = max(worst, rms(d_out[i][:y.size].cpu().numpy(), y))
        assert worst <= RMS_TOL, (launch, worst)
        ls.sync()
        for h, r in zip(hs, refs):
            assert h.state() == r.state(), launch
    ls.close()


def test_fresh_buffers_for_every_launch_keep_the_run_planned_ahead():
    """rsmp_fir_lockstep_rebind_buffers: a service that hands every launch new input and output buffers -- the run planned ahead
    for the launch survives (it starts at the front of the output; the two pointers of its descriptors are patched when it is
    taken over): counts, samples and states as the reference's, and plan-ahead hits from the fourth launch on.  A bind proper
    in between (other capacities would drop the plan) and an appending run after a rebind (which must not survive)."""
    import torch
    dev = torch.device("cuda:0")
    n, k, frames = 14, 40, 128
    specs = sharding.mixed_rate_batch(n, 2, 512)
    hs = [ra.ResamplerFir.new_from_hz(2, s.in_hz, s.out_hz, ra.Latency.Sample64, ra.Attenuation.Db90) for s in specs]
    kind = o.CONVOLVE_AVX_FMA if o.have_avx_fma() else o.CONVOLVE_SCALAR
    refs = [o.OracleFir(2, s.in_hz, s.out_hz, 128, 90, kind) for s in specs]
    caps = [h.buffer_size_output() for h in hs]
    rng = np.random.default_rng(3)
    ls = ra.FirLockstep(hs, frames)
    st = torch.cuda.Stream()
    hits0 = 0
    for launch in range(9):
        xs = [(rng.random(2 * k * frames, dtype=np.float32) * 2 - 1).astype(np.float32) for _ in range(n)]
        d_in = [torch.from_numpy(x).to(dev) for x in xs]
        d_out = [torch.zeros(k * c, device=dev) for c in caps]
        if launch in (0, 6):
            ls.bind_caps(d_in, d_out, caps)
        else:
            ls.rebind(d_in, d_out, st.cuda_stream)
        if launch == 4:
            hits0 = ls.stats()["plan_ahead_hits"]
        ls.run(k, frames, 0, append=(launch == 7), stream=st.cuda_stream)
        st.synchronize()
        cons, prod = ls.run_counts()
        for i, r in enumerate(refs):
            out = np.zeros(caps[i], np.float32)
            ys = []
            for c in range(k):
                rc, cc, pp = r.resample(xs[i][c * 2 * frames:(c + 1) * 2 * frames], out)
                assert rc == 0 and (cc, pp) == (int(cons[c][i]), int(prod[c][i])), (launch, i, c)
                ys.append(out[:pp].copy())
            want = np.concatenate(ys)
            assert rms(d_out[i][:want.size].cpu().numpy(), want) <= RMS_TOL, (launch, i)
        if launch == 5 and os.environ.get("RSMP_LS_AHEAD") != "0":   # (tests/test_knobs_gpu.py runs this without the plan stream as well)
            assert ls.stats()["plan_ahead_hits"] >= hits0 + 2, (hits0, ls.stats())   # (launches 4 and 5: planned ahead, buffers new)
    ls.sync()
    for h, r in zip(hs, refs):
        assert h.state() == r.state()
    ls.close()


def test_bulk_calls_a_stream_cannot_accept_whole_are_refused():
    """The driver loop offers a call's remainder again when the call accepts less than its offer (resample/src/main.rs:
    226-254); a run's calls read at fixed offsets -- so rsmp_fir_lockstep_run_bulk takes only calls every stream accepts whole
    (a stream buffers at most 4096 frames, resampler_fir.rs:18, and keeps up to taps + 1 between calls): anything longer
    is refused, not resampled with frames dropped (found by tools/fuzz_bulk.py through FirBatch's routing, round 6)."""
    import torch
    dev = torch.device("cuda:0")
    hs = [ra.ResamplerFir.new_from_hz(1, 48000, 48000, ra.Latency.Sample8, ra.Attenuation.Db120) for _ in range(2)]
    x = torch.zeros(60000, device=dev)
    caps = [h.buffer_size_output() for h in hs]
    ls = ra.FirLockstep(hs, 4096)
    ls.bind_caps([x, x.clone()], [torch.zeros(70000, device=dev) for _ in hs], caps)
    with pytest.raises(ra.ResampleError):
        ls.run_bulk(60000, 4090)
    ls.run_bulk(60000, 2048)   # (what fits goes through)
    ls.sync()
    ls.close()


@pytest.mark.parametrize("total,chunk", [(20000, 256), (16384, 512), (9999, 300)])
def test_bulk_batch_in_distinct_states_planned_on_the_device(total, chunk):
    """64 streams of six rate pairs in 64 different states (each has already run a different,
    ragged amount through its own handle) take a whole buffer each through rsmp_fir_lockstep_run_bulk -- the reference's
    driver loop (resample/src/main.rs:226-254: calls of `chunk` frames, the last one shorter) planned ON THE DEVICE,
    nothing replayed on the host: every call's (consumed, produced) of every stream, the samples (1e-6 RMS against the
    oracle's AVX+FMA path) and the end states (bit for bit)."""
    import torch
    dev = torch.device("cuda:0")
    n = 64
    specs = sharding.mixed_rate_batch(n, 2, 512)
    hs = [ra.ResamplerFir.new_from_hz(2, s.in_hz, s.out_hz, ra.Latency.Sample64, ra.Attenuation.Db90) for s in specs]
    kind = o.CONVOLVE_AVX_FMA if o.have_avx_fma() else o.CONVOLVE_SCALAR
    refs = [o.OracleFir(2, s.in_hz, s.out_hz, 128, 90, kind) for s in specs]
    rng = np.random.default_rng(total)
    for i, (h, r) in enumerate(zip(hs, refs)):   # distinct states: 64 + 37 i frames in ragged calls
        x = (rng.random(2 * (64 + 37 * i), dtype=np.float32) * 2 - 1).astype(np.float32)
        og, orr = np.zeros(h.buffer_size_output(), np.float32), np.zeros(r.buffer_size_output(), np.float32)
        off = 0
        while off < x.size:
            cg, pg = h.resample(x[off:off + 2 * 211], og)
            rc, cr, pr = r.resample(x[off:off + 2 * 211], orr)
            assert rc == 0 and (cg, pg) == (cr, pr)
            off += cg
    assert len({h.state() for h in hs}) > n // 2
    xs = [(rng.random(2 * total, dtype=np.float32) * 2 - 1).astype(np.float32) for _ in range(n)]
    d_in = [torch.from_numpy(x).to(dev) for x in xs]
    caps = [h.buffer_size_output() for h in hs]
    k = total // chunk
    d_out = [torch.zeros((k + 2) * c, device=dev) for c in caps]
    ls = ra.FirLockstep(hs, 512)
    ls.bind_caps(d_in, d_out, caps)
    ls.run_bulk(total, chunk)
    cons, prod = ls.run_counts()
    tail = total - k * chunk
    if tail:
        ct, pt = ls.counts()
    worst = 0.0
    for i, r in enumerate(refs):
        y, calls = r.resample_all(xs[i], 2 * chunk, max_calls=k + 4)
        assert calls.shape[0] == k + (1 if tail else 0), (i, calls.shape)
        assert (calls[:k, 0] == cons[:, i]).all() and (calls[:k, 1] == prod[:, i]).all(), i
        if tail:
            assert (int(ct[i]), int(pt[i])) == (int(calls[k, 0]), int(calls[k, 1])), i
        worst = max(worst, rms(d_out[i][:y.size].cpu().numpy(), y))
    assert worst <= RMS_TOL, worst
    ls.sync()
    for h, r in zip(hs, refs):
        assert h.state() == r.state()
