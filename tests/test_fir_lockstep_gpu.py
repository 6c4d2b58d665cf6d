"""GPU parity tests of the lock-step batch (rsmp_fir_lockstep_*, BASELINE config 4): per stream and step
exactly one reference resample() call (src/resampler_fir.rs:509-621) -- identical (consumed, produced),
output within 1e-6 RMS of the CPU oracle -- with the streams' state resident in HBM and the control flow
run inside the kernel."""
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


def run_lockstep(specs, steps, frames, seed=0, lat=ra.Latency.Sample64, att=ra.Attenuation.Db90,
                 prefeed=None, in_frames_per_stream=None, x_override=None, append=True, exact=None, relative=False):
    """Runs `steps` lock-step steps of `frames` frames over the streams in `specs` on the GPU and the same
    calls through one OracleFir per stream.  Returns the worst RMS error over the streams; asserts equal
    counts at every step.  prefeed[i]: frames pushed through stream i with ordinary resample() calls
    beforehand, so that the streams are in different states."""
    import torch
    dev = torch.device("cuda:0")
    n = len(specs)
    hs = [ra.ResamplerFir.new_from_hz(s.channels, s.in_hz, s.out_hz, lat, att) for s in specs]
    if exact is not None:   # exact(i): stream i keeps every product in f32 (its own workgroups inside the batch)
        for i, h in enumerate(hs):
            if exact(i):
                h.set_kernel(ra.FirKernel.PeriodicF32)
    kind = o.CONVOLVE_AVX_FMA if o.have_avx_fma() else o.CONVOLVE_SCALAR   # the CPU SIMD path north_star names
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
    xs = []
    for i, s in enumerate(specs):
        if x_override is not None:
            xs.append(x_override[i])
        else:
            xs.append((rng.random(steps * frames * s.channels, dtype=np.float32) * 2 - 1).astype(np.float32))
    d_in = [torch.from_numpy(x).to(dev) for x in xs]
    caps = [h.buffer_size_output() for h in hs]
    d_out = [torch.zeros(steps * c if append else c, device=dev, dtype=torch.float32) for c in caps]
    ls = ra.FirLockstep(hs, frames)
    ls.bind_caps(d_in, d_out, caps)
    d_fr = None
    if in_frames_per_stream is not None:
        d_fr = torch.tensor(in_frames_per_stream, dtype=torch.int32, device=dev)
    ref_out = [[] for _ in range(n)]
    gpu_step_out = [[] for _ in range(n)]
    orr = [np.zeros(r.buffer_size_output(), np.float32) for r in refs]
    for k in range(steps):
        ls.step(frames, k * frames, append=append, d_in_frames=d_fr)
        cons, prod = ls.counts()
        for i, s in enumerate(specs):
            fr = frames if in_frames_per_stream is None else min(frames, in_frames_per_stream[i])
            sl = xs[i][k * frames * s.channels:(k * frames + fr) * s.channels]
            rc, cr, pr = refs[i].resample(sl, orr[i])
            assert rc == 0
            assert (int(cons[i]), int(prod[i])) == (cr, pr), (k, i, (cons[i], prod[i]), (cr, pr))
            ref_out[i].append(orr[i][:pr].copy())
            if not append:
                gpu_step_out[i].append(d_out[i][:pr].cpu().numpy())
    worst = 0.0
    for i in range(n):
        want = np.concatenate(ref_out[i]) if ref_out[i] else np.zeros(0, np.float32)
        got = d_out[i][:want.size].cpu().numpy() if append else np.concatenate(gpu_step_out[i])
        e = rms(got, want)
        if relative:   # relative to the stream's own level
            e /= max(float(np.sqrt(np.mean(want.astype(np.float64) ** 2))), 1e-300)
        if e > RMS_TOL:
            bad = np.flatnonzero(~(np.abs(got.astype(np.float64) - want) <= 1e-4))
            print(f"lockstep: stream {i} {specs[i]} rms {e:.3e} bad {bad.size}: {bad[:8]} got {got[bad[:4]]} "
                  f"want {want[bad[:4]]}")
        worst = max(worst, e)
    return worst, ls, hs, refs


def test_streams_of_very_different_levels_keep_their_relative_precision():
    """Streams of one batch (and of one workgroup: up to five share an image) at full scale, 2^-10, 2^-17, 2^-24, 1e-30
    and 3e4, one of them dropping from full scale to 2^-20 halfway: every stream within 1e-6 RMS RELATIVE to its own
    level.  The split image is block floating point per column (fir_lockstep.hip, `colpeak`)."""
    specs = sharding.mixed_rate_batch(42, 2, 512)
    levels = [1.0, 2.0 ** -10, 2.0 ** -17, 2.0 ** -24, 1e-30, 3e4, 1.0]
    steps, frames = 6, 512
    rng = np.random.default_rng(123)
    xs = []
    for i, sp in enumerate(specs):
        x = (rng.random(steps * frames * 2, dtype=np.float32) * 2 - 1).astype(np.float32) * np.float32(levels[(i // 6) % 7])
        if (i // 6) % 7 == 6:
            x[steps * frames:] *= np.float32(2.0 ** -20)
        xs.append(x.astype(np.float32))
    worst, ls, hs, refs = run_lockstep(specs, steps=steps, frames=frames, x_override=xs, relative=True)
    assert worst <= RMS_TOL, worst
    # (3e4 overflows nothing: the scale follows the stream's peak; no scale is derived from a peak of 2^11 and above --
    # kLsPeakMax -- so samples of 2^13 and above overflow the planes and go to the reference form)
    ls.close()


def test_c4_shape_1024_streams_six_pairs_16_steps():
    # BASELINE config 4: 1024 streams, stream i = ordered pair i mod 6 of the 44.1k / 48k / 96k
    # conversions, 2 ch, 128 taps, Db90, lock-step steps of 512 frames on carried state.
    specs = sharding.mixed_rate_batch(1024, 2, 512)
    worst, ls, hs, refs = run_lockstep(specs, steps=16, frames=512, seed=4)
    assert worst <= RMS_TOL, worst
    assert not ls.status().any()
    # the device state written back into the handles equals the oracle's state machine, bit for bit
    ls.sync()
    for h, r in zip(hs, refs):
        assert h.state() == r.state()
    x = synth.fast_noise(2 * 300, seed=9)
    ls.close()
    for i in (0, 1, 2, 3, 4, 5, 1023):   # the handles carry on through the ordinary API
        g_out = np.zeros(hs[i].buffer_size_output(), np.float32)
        r_out = np.zeros(refs[i].buffer_size_output(), np.float32)
        cg, pg = hs[i].resample(x, g_out)
        rc, cr, pr = refs[i].resample(x, r_out)
        assert rc == 0 and (cg, pg) == (cr, pr)
        assert rms(g_out[:pg], r_out[:pr]) <= RMS_TOL


@pytest.mark.parametrize("which", ["all-exact", "mixed"])
def test_exact_f32_products_on_request_and_mixed_batches(which):
    # two-channel streams run on the fp16 matrix cores with split operands by default; a stream set to
    # RSMP_FIR_KERNEL_PERIODIC_F32 keeps exact f32 products, also next to split streams of the same rate pair
    specs = sharding.mixed_rate_batch(60, 2, 512)
    exact = (lambda i: True) if which == "all-exact" else (lambda i: i % 5 < 2)
    worst, ls, hs, refs = run_lockstep(specs, steps=5, frames=512, seed=21, exact=exact, prefeed=[(53 * i) % 900 for i in range(60)])
    assert worst <= RMS_TOL, worst
    assert not ls.status().any()
    if which == "all-exact":
        assert ls.split_workgroups() == 0
    else:
        assert 0 < ls.split_workgroups() < ls.workgroups()
    ls.sync()
    for h, r in zip(hs, refs):
        assert h.state() == r.state()


def test_default_two_channel_batch_runs_split_and_other_channel_counts_do_not():
    import torch
    dev = torch.device("cuda:0")
    for ch, want_split in ((2, True), (1, False), (6, False)):
        hs = [ra.ResamplerFir.new_from_hz(ch, 44100, 48000, ra.Latency.Sample64, ra.Attenuation.Db90) for _ in range(7)]
        ls = ra.FirLockstep(hs, 256)
        assert (ls.split_workgroups() == ls.workgroups()) == want_split, (ch, ls.split_workgroups(), ls.workgroups())
        ls.close()


def test_high_ratio_steps_with_more_columns_than_one_image():
    # 8 kHz -> 192 kHz: a 200-frame step of one stream spans 54 super periods = 54 columns; the split variant
    # holds 16 per workgroup, so this geometry keeps the f32 layout (found by tools/fuzz_lockstep.py)
    specs = [sharding.StreamSpec(2, 8000, 192000, 16, 200) for _ in range(5)]
    worst, ls, _, _ = run_lockstep(specs, steps=5, frames=200, seed=31, lat=ra.Latency.Sample8)
    assert worst <= RMS_TOL, worst
    assert ls.split_workgroups() == 0


def test_unaligned_input_pointers():
    # the split variant reads frames with 8-byte loads only when every input pointer allows it
    import torch
    dev = torch.device("cuda:0")
    specs = sharding.mixed_rate_batch(12, 2, 300)
    steps, frames = 4, 300
    hs = [ra.ResamplerFir.new_from_hz(2, s.in_hz, s.out_hz, ra.Latency.Sample64, ra.Attenuation.Db90) for s in specs]
    refs = [o.OracleFir(2, s.in_hz, s.out_hz, 128, 90) for s in specs]
    xs = [synth.fast_noise(steps * frames * 2, seed=40 + i) for i in range(12)]
    store = [torch.zeros(steps * frames * 2 + 1, device=dev) for _ in range(12)]
    d_in = []
    for i in range(12):
        store[i][1:] = torch.from_numpy(xs[i]).to(dev)
        d_in.append(store[i][1:])                     # 4 bytes off an 8-byte boundary
    assert all(t.data_ptr() % 8 == 4 for t in d_in)
    caps = [h.buffer_size_output() for h in hs]
    d_out = [torch.zeros(steps * c, device=dev) for c in caps]
    ls = ra.FirLockstep(hs, frames)
    ls.bind_caps(d_in, d_out, caps)
    for k in range(steps):
        ls.step(frames, k * frames, append=True)
    ls.counts()
    for i in range(12):
        y, _ = refs[i].resample_all(xs[i], frames * 2)
        assert rms(d_out[i][:y.size].cpu().numpy(), y) <= RMS_TOL, i


def test_streams_in_different_states():
    # every stream has been running for a different time: different positions, buffered frames, phases
    specs = sharding.mixed_rate_batch(96, 2, 512)
    prefeed = [(37 * i) % 1500 for i in range(96)]
    worst, ls, _, _ = run_lockstep(specs, steps=6, frames=512, seed=5, prefeed=prefeed)
    assert worst <= RMS_TOL, worst
    assert not ls.status().any()


@pytest.mark.parametrize("channels", [1, 3, 8])
def test_channel_counts(channels):
    pairs = [(44100, 48000), (96000, 44100), (48000, 96000)]
    specs = [sharding.StreamSpec(channels, *pairs[i % 3], 128, 256) for i in range(9)]
    worst, ls, _, _ = run_lockstep(specs, steps=5, frames=256, seed=6 + channels)
    assert worst <= RMS_TOL, worst


@pytest.mark.parametrize("lat", [ra.Latency.Sample8, ra.Latency.Sample16, ra.Latency.Sample32])
def test_tap_counts_and_other_rates(lat):
    pairs = [(22050, 48000), (48000, 32000), (32000, 48000), (16000, 44100), (192000, 48000), (48000, 192000)]
    specs = [sharding.StreamSpec(2, *pairs[i % 6], lat.taps(), 200) for i in range(12)]
    worst, ls, _, _ = run_lockstep(specs, steps=7, frames=200, seed=11, lat=lat, att=ra.Attenuation.Db60)
    assert worst <= RMS_TOL, worst


def test_irrational_and_huge_ratios_use_the_reference_form():
    # no usable period: every output is evaluated in the reference's two-row form inside the same kernel
    specs = [sharding.StreamSpec(2, 44100, 44101, 128, 384), sharding.StreamSpec(2, 48000, 47999, 128, 384),
             sharding.StreamSpec(1, 12345, 54321, 128, 384), sharding.StreamSpec(2, 44100, 48000, 128, 384)]
    worst, ls, _, _ = run_lockstep(specs, steps=6, frames=384, seed=12)
    assert worst <= RMS_TOL, worst


def test_per_stream_step_sizes_and_call_semantics():
    # ragged steps (a device array of frames per stream, including empty ones), output written at the
    # start of `out` every step like the reference call
    specs = sharding.mixed_rate_batch(18, 2, 512)
    fr = [0, 1, 17, 128, 511, 512] * 3
    worst, ls, _, _ = run_lockstep(specs, steps=5, frames=512, seed=13, in_frames_per_stream=fr, append=False)
    assert worst <= RMS_TOL, worst


def test_non_finite_samples_match_the_reference():
    # Inf / NaN at a stream start, in the middle of a period, at a step edge, in one channel and in both:
    # the outputs the reference leaves finite stay finite and equal, the others are non-finite in both.
    import torch
    specs = sharding.mixed_rate_batch(12, 2, 512)
    steps, frames = 4, 512
    rng = np.random.default_rng(21)
    xs = [(rng.random(steps * frames * 2, dtype=np.float32) * 2 - 1).astype(np.float32) for _ in specs]
    xs[0][0] = np.inf                       # first sample of the stream, channel 0
    xs[1][2 * 700 + 1] = -np.inf            # mid stream, channel 1
    xs[2][2 * 511] = np.nan                 # last frame of step 0
    xs[2][2 * 511 + 1] = np.nan
    xs[3][2 * 512] = np.inf                 # first frame of step 1
    xs[4][2 * 1000] = np.inf
    xs[4][2 * 1003] = -np.inf               # +inf and -inf inside one window
    xs[5][2 * 1500 + 1] = np.nan
    dev = torch.device("cuda:0")
    hs = [ra.ResamplerFir.new_from_hz(2, s.in_hz, s.out_hz, ra.Latency.Sample64, ra.Attenuation.Db90) for s in specs]
    refs = [o.OracleFir(2, s.in_hz, s.out_hz, 128, 90, o.CONVOLVE_AVX_FMA if o.have_avx_fma() else o.CONVOLVE_SCALAR)
            for s in specs]
    caps = [h.buffer_size_output() for h in hs]
    d_in = [torch.from_numpy(x).to(dev) for x in xs]
    d_out = [torch.zeros(steps * c, device=dev) for c in caps]
    ls = ra.FirLockstep(hs, frames)
    ls.bind_caps(d_in, d_out, caps)
    want = [[] for _ in specs]
    orr = [np.zeros(c, np.float32) for c in caps]
    for k in range(steps):
        ls.step(frames, k * frames, append=True)
        cons, prod = ls.counts()
        for i in range(len(specs)):
            rc, cr, pr = refs[i].resample(xs[i][k * frames * 2:(k + 1) * frames * 2], orr[i])
            assert rc == 0 and (int(cons[i]), int(prod[i])) == (cr, pr)
            want[i].append(orr[i][:pr].copy())
    for i in range(len(specs)):
        w = np.concatenate(want[i])
        g = d_out[i][:w.size].cpu().numpy()
        fin = np.isfinite(w)
        assert np.array_equal(np.isfinite(g), fin), (i, np.flatnonzero(np.isfinite(g) != fin)[:10])
        assert np.array_equal(np.isnan(g), np.isnan(w)), i
        inf = np.isinf(w)
        assert np.array_equal(g[inf], w[inf]), i          # same signed infinities
        assert rms(g[fin], w[fin]) <= RMS_TOL
        if i <= 5:
            assert (~fin).any()
    st = ls.status()
    assert all(st[i] & 2 for i in range(6)) and not any(st[6:])


def test_reset_and_rebind():
    import torch
    specs = sharding.mixed_rate_batch(12, 2, 256)
    worst, ls, hs, refs = run_lockstep(specs, steps=3, frames=256, seed=31)
    assert worst <= RMS_TOL
    ls.reset()
    for r in refs:
        r.reset()
    dev = torch.device("cuda:0")
    x = synth.fast_noise(2 * 256, seed=32)
    d_in = [torch.from_numpy(x).to(dev) for _ in specs]
    d_out = [torch.zeros(h.buffer_size_output(), device=dev) for h in hs]
    ls.bind(d_in, d_out)
    ls.step(256)
    cons, prod = ls.counts()
    for i, r in enumerate(refs):
        out = np.zeros(r.buffer_size_output(), np.float32)
        rc, cr, pr = r.resample(x, out)
        assert rc == 0 and (int(cons[i]), int(prod[i])) == (cr, pr)
        assert rms(d_out[i][:pr].cpu().numpy(), out[:pr]) <= RMS_TOL


@pytest.mark.parametrize("steps_a,steps_b", [(1, 2), (3, 1), (4, 3)])
def test_history_buffers_alternate_across_binds_batches_and_plain_calls(steps_a, steps_b):
    """A stream's buffered frames move between its two history buffers every step.  Whatever the parity of the
    step count: a rebind in the middle, the end of the batch, ordinary resample() calls on the handles and a
    second batch over the same handles all continue the same streams (checked against one oracle per stream)."""
    import torch
    dev = torch.device("cuda:0")
    specs = sharding.mixed_rate_batch(12, 2, 192) + [sharding.StreamSpec(3, 48000, 44100, 128, 192)]
    frames = 192
    hs = [ra.ResamplerFir.new_from_hz(s.channels, s.in_hz, s.out_hz, ra.Latency.Sample64, ra.Attenuation.Db90) for s in specs]
    refs = [o.OracleFir(s.channels, s.in_hz, s.out_hz, 128, 90) for s in specs]
    caps = [h.buffer_size_output() for h in hs]
    rng = np.random.default_rng(77)
    ref_buf = [np.zeros(c, np.float32) for c in caps]

    def batch_steps(ls, n_steps):
        xs = [(rng.random(n_steps * frames * s.channels, dtype=np.float32) * 2 - 1).astype(np.float32) for s in specs]
        d_in = [torch.from_numpy(x).to(dev) for x in xs]
        d_out = [torch.zeros(n_steps * c, device=dev) for c in caps]
        ls.bind_caps(d_in, d_out, caps)
        for k in range(n_steps):
            ls.step(frames, k * frames, append=True)
        ls.counts()
        for i, s in enumerate(specs):
            want = []
            for k in range(n_steps):
                rc, cr, pr = refs[i].resample(xs[i][k * frames * s.channels:(k + 1) * frames * s.channels], ref_buf[i])
                assert rc == 0
                want.append(ref_buf[i][:pr].copy())
            want = np.concatenate(want)
            assert rms(d_out[i][:want.size].cpu().numpy(), want) <= RMS_TOL, i

    def plain_calls():
        for i, s in enumerate(specs):
            x = (rng.random(100 * s.channels, dtype=np.float32) * 2 - 1).astype(np.float32)
            g_out = np.zeros(caps[i], np.float32)
            cg, pg = hs[i].resample(x, g_out)
            rc, cr, pr = refs[i].resample(x, ref_buf[i])
            assert rc == 0 and (cg, pg) == (cr, pr)
            assert rms(g_out[:pg], ref_buf[i][:pr]) <= RMS_TOL, i

    ls = ra.FirLockstep(hs, frames)
    batch_steps(ls, steps_a)
    batch_steps(ls, steps_b)          # rebind inside the batch
    ls.close()
    for h, r in zip(hs, refs):
        assert h.state() == r.state()
    plain_calls()
    ls2 = ra.FirLockstep(hs, frames)   # a second batch over the same handles
    batch_steps(ls2, steps_a)
    ls2.close()
    plain_calls()


def test_rejects_small_output_and_busy_streams():
    import torch
    dev = torch.device("cuda:0")
    h = ra.ResamplerFir.new_from_hz(2, 44100, 48000, ra.Latency.Sample64, ra.Attenuation.Db90)
    ls = ra.FirLockstep([h], 512)
    d_in = [torch.zeros(1024, device=dev)]
    with pytest.raises(ra.InvalidOutputBufferSize):
        ls.bind(d_in, [torch.zeros(64, device=dev)])
    with pytest.raises(ra.ResampleError):
        ls.step(512)            # nothing bound
    ls.bind(d_in, [torch.zeros(h.buffer_size_output(), device=dev)])
    with pytest.raises(ra.InvalidInputBufferSize):
        ls.step(513)            # more than the batch was created for
    with pytest.raises(ra.ResampleError):
        ra.FirLockstep([h, h], 512)


def _feed_steps(feed, ls, specs, caps, refs, dev, stream_arg, steps=5, frames=512):
    import torch
    n = len(specs)
    stage_out = torch.zeros(sum(caps), device=dev)
    rng = np.random.default_rng(77)
    for k in range(steps):
        x = (rng.random(n * frames * 2, dtype=np.float32) * 2 - 1).astype(np.float32)
        d_x = torch.from_numpy(x).to(dev, non_blocking=True)
        # scatter -> step -> gather: nothing but the stream orders them (no host sync in between)
        feed.scatter(d_x)
        ls.step(frames, 0, stream=stream_arg)
        feed.gather(stage_out)
        cons, prod = ls.counts()
        torch.cuda.current_stream().synchronize()
        got = stage_out.cpu().numpy()
        for i in range(n):
            out = np.zeros(caps[i], np.float32)
            rc, c, p = refs[i].resample(x[i * frames * 2:(i + 1) * frames * 2], out)
            assert rc == 0 and (int(cons[i]), int(prod[i])) == (c, p)
            assert rms(got[feed.out_off[i]:feed.out_off[i] + p], out[:p]) <= RMS_TOL


def _feed_fixture(n=30, frames=512):
    specs = sharding.mixed_rate_batch(n, 2, frames)
    parts = sharding.partition([s.work() for s in specs], 1)
    caps = [sharding.buffer_size_output(s) for s in specs]
    hs = [ra.ResamplerFir.new_from_hz(2, s.in_hz, s.out_hz, ra.Latency.Sample64, ra.Attenuation.Db90) for s in specs]
    assert caps == [h.buffer_size_output() for h in hs]
    refs = [o.OracleFir(2, s.in_hz, s.out_hz, 128, 90) for s in specs]
    return specs, parts, caps, hs, refs


@pytest.mark.parametrize("side_stream", [False, True])
def test_sharded_step_feed_on_device(side_stream):
    # The per-rank device path of the multi-GPU form (bench.py --config c4 --feed rccl) at world size 1:
    # a step's chunks arrive through sharding.StepFeed's exchange buffers, the streams are bound straight
    # to their slices of them, the outputs leave through the gather side.  (World 2 moves the same
    # buffers over gloo in tests/test_sharding_gloo.py.)  The step runs on torch's CURRENT stream -- the
    # default one (passed as RSMP_STREAM_LEGACY: a NULL stream argument would mean the batch's own
    # non-blocking stream, ordered against neither the scatter nor the gather) or a side stream.
    import torch
    dev = torch.device("cuda:0")
    n, frames = 30, 512
    specs, parts, caps, hs, refs = _feed_fixture(n, frames)
    feed = sharding.StepFeed(None, 0, 1, parts, [frames * 2] * n, caps, dev)
    ls = ra.FirLockstep(hs, frames)
    ls.bind_caps([feed.local_in_view(i) for i in range(n)], [feed.local_out_view(i) for i in range(n)], caps)
    torch.cuda.synchronize()
    if side_stream:
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            assert ra.torch_stream() == st.cuda_stream != 0
            _feed_steps(feed, ls, specs, caps, refs, dev, ra.torch_stream())
    else:
        assert ra.torch_stream() == ra.STREAM_LEGACY
        _feed_steps(feed, ls, specs, caps, refs, dev, ra.torch_stream())
    ls.close()


def test_sharded_step_feed_through_rccl_at_world_one():
    # The same step with the exchange really going through RCCL (backend "nccl"): a world of one rank whose
    # GPU sends its shard to itself and receives it in the same group (StepFeed loopback), on a side stream.
    # What this executes on hardware: RCCL init, the grouped ncclSend / ncclRecv launch of batch_isend_irecv and
    # the stream ordering scatter -> fir_lockstep_kernel -> gather.  (More than one GPU: the driver's scaling run.)
    import socket
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    dev = torch.device("cuda:0")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=dev)
    try:
        n, frames = 30, 512
        specs, parts, caps, hs, refs = _feed_fixture(n, frames)
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            feed = sharding.StepFeed(dist, 0, 1, parts, [frames * 2] * n, caps, dev, loopback=True)
            ls = ra.FirLockstep(hs, frames)
            ls.bind_caps([feed.local_in_view(i) for i in range(n)], [feed.local_out_view(i) for i in range(n)], caps)
            torch.cuda.synchronize()
            _feed_steps(feed, ls, specs, caps, refs, dev, ra.torch_stream(), steps=4)
            ls.close()
    finally:
        dist.destroy_process_group()
