"""FFT path: host-side planning parity (CPU) and GPU parity of ResamplerFft against the oracle.
Gate (BASELINE.json north_star): output within 1e-6 RMS of the CPU path."""
import itertools

import numpy as np
import pytest

import resampler_amd as ra
from oracle import pyoracle as o
from resampler_amd import synth

RMS_TOL = 1e-6
RATES = [22050, 16000, 32000, 44100, 48000, 88200, 96000, 176400, 192000, 384000]


def rms(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


def sr(hz):
    return ra.SampleRate(RATES.index(hz))


# ---- CPU: planning ------------------------------------------------------------------------------
def test_plan_matches_oracle_for_every_rate_pair():
    for i, out in itertools.product(RATES, RATES):
        fi, fo, a, b = o.fft_plan(i, out)
        mi, mo, fwd, inv = ra.fft_plan_sizes(i, out)
        assert (mi, mo) == (fi, fo)
        assert fwd == o.OracleRfft(a + [2]).stage_factors()
        assert inv == o.OracleRfft(b + [2], inverse=True).stage_factors()
    assert ra.fft_plan_sizes(44100, 48000) == (1176, 1280, [3, 7, 7, 8], [4, 5, 8, 8])
    with pytest.raises(ra.ResampleError):
        ra.fft_plan_sizes(24000, 48000)


# ---- GPU ----------------------------------------------------------------------------------------
def consistent(ch, in_hz, out_hz):
    """Rate pairs / channel counts where the reference's per-channel scratch regions do not collide
    (SURVEY 7.3 item 6): channel c's block lives at c*fft_in*(ch+1) in a scratch of
    fft_out*(ch+1)*ch values."""
    fi, fo, _, _ = o.fft_plan(in_hz, out_hz)
    stride = fi * (ch + 1)
    return ch == 1 or (fo <= stride and (ch - 1) * stride + fo <= fo * (ch + 1) * ch)


@pytest.mark.gpu
@pytest.mark.parametrize("ch,in_hz,out_hz", [
    (2, 44100, 48000),      # BASELINE config 3
    (2, 48000, 44100),
    (1, 48000, 32000),
    (1, 32000, 48000),
    (1, 96000, 48000),
    (2, 48000, 96000),
    (1, 16000, 44100),
    (1, 22050, 16000),
    (3, 44100, 48000),
    (1, 44100, 44100),
    (1, 22050, 48000),
    (2, 88200, 96000),
    # blocks of 7056 .. 12288 frames: one transform buffer, in place (fft_ola_big_kernel)
    (1, 44100, 384000),
    (1, 384000, 44100),
    (1, 16000, 384000),
    (1, 176400, 384000),
    (2, 176400, 384000),
    (1, 32000, 176400),
])
def test_per_call_resample_matches_oracle(ch, in_hz, out_hz):
    assert consistent(ch, in_hz, out_hz)
    g = ra.ResamplerFft.new(ch, sr(in_hz), sr(out_hz))
    r = o.OracleFft(ch, in_hz, out_hz)
    n_in, n_out = g.chunk_size_input(), g.chunk_size_output()
    assert (n_in, n_out, g.delay()) == (r.chunk_size_input(), r.chunk_size_output(), r.delay())
    blocks = 6
    x = synth.sweep(blocks * n_in // ch, ch, float(in_hz))
    og = np.zeros(n_out, np.float32)
    orr = np.zeros(n_out, np.float32)
    for b in range(blocks):
        g.resample(x[b * n_in:(b + 1) * n_in], og)
        assert r.resample(x[b * n_in:(b + 1) * n_in], orr) == 0
        assert rms(og, orr) <= RMS_TOL, b


@pytest.mark.gpu
def test_error_codes_and_oversized_buffers():
    g = ra.ResamplerFft.new(2, ra.SampleRate.Hz44100, ra.SampleRate.Hz48000)
    r = o.OracleFft(2, 44100, 48000)
    out = np.zeros(2560, np.float32)
    with pytest.raises(ra.InvalidInputBufferSize):
        g.resample(np.zeros(2351, np.float32), out)
    with pytest.raises(ra.InvalidOutputBufferSize):
        g.resample(np.zeros(2352, np.float32), out[:2559])
    # `>=` is enough; extra values are ignored (resampler_fft.rs:186-192)
    x = synth.fast_noise(3000, seed=4)
    og = np.zeros(3000, np.float32)
    orr = np.zeros(3000, np.float32)
    g.resample(x, og)
    assert r.resample(x, orr) == 0
    assert rms(og[:2560], orr[:2560]) <= RMS_TOL and not og[2560:].any()
    with pytest.raises((ValueError, ra.ResampleError)):
        ra.ResamplerFft(2, 99, 4)
    assert not ra.lib().rsmp_fft_new(2, 99, 4, 0)


@pytest.mark.gpu
def test_one_handle_across_streams_and_twice_in_a_batch():
    """A handle's calls are ordered whatever stream each is enqueued on (`&mut self`, resampler_fft.rs:182):
    call k + 1 reads the overlap rows call k writes.  A handle listed twice in one batch is refused."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    g = ra.ResamplerFft.new(2, ra.SampleRate.Hz44100, ra.SampleRate.Hz48000)
    r = o.OracleFft(2, 44100, 48000)
    n_in, n_out = g.chunk_size_input(), g.chunk_size_output()
    blocks, calls = 40, 6
    x = synth.fast_noise(calls * blocks * n_in, seed=21)
    d_x = torch.from_numpy(x).to(dev)
    d_y = torch.zeros(calls * blocks * n_out, device=dev)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    for k in range(calls):   # no host sync between the calls; the stream changes every time
        g.resample_bulk_device(d_x[k * blocks * n_in:(k + 1) * blocks * n_in],
                               d_y[k * blocks * n_out:(k + 1) * blocks * n_out], blocks, streams[k % 3].cuda_stream)
    torch.cuda.synchronize()
    want = np.zeros(calls * blocks * n_out, np.float32)
    for b in range(calls * blocks):
        assert r.resample(x[b * n_in:(b + 1) * n_in], want[b * n_out:(b + 1) * n_out]) == 0
    assert rms(d_y.cpu().numpy(), want) <= RMS_TOL
    batch = ra.FftBatch([g, g])
    batch.bind([d_x, d_x], [d_y, d_y], [1, 1])
    with pytest.raises(ra.ResampleError):
        batch.resample_bulk_device()


@pytest.mark.gpu
@pytest.mark.parametrize("ch,in_hz,out_hz,blocks", [(2, 44100, 48000, 37), (1, 48000, 44100, 50),
                                                    (2, 48000, 96000, 20)])
def test_bulk_equals_consecutive_calls(ch, in_hz, out_hz, blocks):
    g = ra.ResamplerFft.new(ch, sr(in_hz), sr(out_hz))
    r = o.OracleFft(ch, in_hz, out_hz)
    n_in, n_out = g.chunk_size_input(), g.chunk_size_output()
    x = synth.fast_noise(blocks * n_in, seed=blocks)
    # a few per-call chunks first so the bulk launch starts from a non-zero overlap
    og = np.zeros(n_out, np.float32)
    ref = np.zeros((blocks, n_out), np.float32)
    for b in range(blocks):
        assert r.resample(x[b * n_in:(b + 1) * n_in], ref[b]) == 0
    for b in range(3):
        g.resample(x[b * n_in:(b + 1) * n_in], og)
        assert rms(og, ref[b]) <= RMS_TOL
    y = g.resample_bulk(x[3 * n_in:], blocks - 3)     # > kFftRun blocks: exercises the halo recompute
    assert rms(y, ref[3:].reshape(-1)) <= RMS_TOL
    assert float(np.max(np.abs(y - ref[3:].reshape(-1)))) < 1e-5


def _oracle_blocks(in_hz, out_hz, x, blocks):
    r = o.OracleFft(2, in_hz, out_hz)
    n_in, n_out = r.chunk_size_input(), r.chunk_size_output()
    ref = np.zeros((blocks, n_out), np.float32)
    for b in range(blocks):
        assert r.resample(x[b * n_in:(b + 1) * n_in], ref[b]) == 0
    return ref.reshape(-1)


@pytest.mark.gpu
@pytest.mark.parametrize("in_hz,out_hz", [(44100, 48000), (48000, 44100)])
@pytest.mark.parametrize("quiet_exp", [12, 20, 30])
def test_channels_of_a_two_channel_stream_at_very_different_levels(in_hz, out_hz, quiet_exp):
    """The reference transforms every channel on its own (resampler_fft.rs:182-240), so a quiet channel's error is relative to
    ITS level.  The two-channel kernel (fft_pair.hip) runs both channels through the same butterflies as one complex signal:
    every channel of every block is brought to a common level first, and each must stay within the gate of its own RMS."""
    g = ra.ResamplerFft.new(2, sr(in_hz), sr(out_hz))
    n_in = g.chunk_size_input()
    blocks = 40
    x = synth.fast_noise(blocks * n_in, seed=quiet_exp).copy()
    for quiet in (0, 1):
        xx = x.copy()
        xx[quiet::2] *= np.float32(2.0 ** -quiet_exp)
        # (the quiet channel wanders through the levels: a block that is louder than its neighbours, a silent block)
        fr = n_in // 2
        xx[quiet + 2 * 7 * fr:2 * 8 * fr:2] *= np.float32(2.0 ** 9)
        xx[quiet + 2 * 11 * fr:2 * 12 * fr:2] = 0.0
        ref = _oracle_blocks(in_hz, out_hz, xx, blocks)
        y = g.resample_bulk(xx, blocks)
        g = ra.ResamplerFft.new(2, sr(in_hz), sr(out_hz))
        for c in (0, 1):
            level = float(np.sqrt(np.mean(ref[c::2].astype(np.float64) ** 2)))
            assert rms(y[c::2], ref[c::2]) <= RMS_TOL * level, (quiet, c, rms(y[c::2], ref[c::2]) / level)


@pytest.mark.gpu
@pytest.mark.parametrize("in_hz,out_hz", [(44100, 48000), (48000, 44100)])
@pytest.mark.parametrize("level_l,level_r", [(1e-30, 1e-30), (1e-30, 1.0), (3e4, 1e-12), (1e-30, 1e3), (1e15, 1e15)])   # (below ~1e-33 the REFERENCE works in denormals: X H is 1e-5 of the input)
def test_two_channel_streams_at_the_ends_of_the_f32_range(in_hz, out_hz, level_l, level_r):
    """f32 arithmetic does not care where a signal sits in its range, and the reference's per-channel transforms inherit
    that; the two-channel kernel's level estimate (an energy: squares) must not turn a tiny channel into a silent one or a
    large one into a dead one."""
    g = ra.ResamplerFft.new(2, sr(in_hz), sr(out_hz))
    n_in = g.chunk_size_input()
    blocks = 9
    x = synth.fast_noise(blocks * n_in, seed=3).copy()
    x[0::2] *= np.float32(level_l)
    x[1::2] *= np.float32(level_r)
    with np.errstate(under="ignore"):
        ref = _oracle_blocks(in_hz, out_hz, x, blocks)
        y = g.resample_bulk(x, blocks)
    for c in (0, 1):
        level = float(np.sqrt(np.mean(ref[c::2].astype(np.float64) ** 2)))
        assert level > 0.0 and rms(y[c::2], ref[c::2]) <= RMS_TOL * level, (c, level, rms(y[c::2], ref[c::2]) / level)


@pytest.mark.gpu
@pytest.mark.parametrize("in_hz,out_hz", [(44100, 48000), (48000, 44100)])
def test_a_silent_channel_stays_silent_and_a_nan_stays_in_its_channel(in_hz, out_hz):
    g = ra.ResamplerFft.new(2, sr(in_hz), sr(out_hz))
    n_in, n_out = g.chunk_size_input(), g.chunk_size_output()
    blocks = 12
    x = synth.fast_noise(blocks * n_in, seed=5).copy()
    x[1::2] = 0.0
    y = g.resample_bulk(x, blocks)
    assert np.all(y[1::2] == 0.0)                                   # zeros in, zeros out: nothing of channel 0 leaks over
    ref = _oracle_blocks(in_hz, out_hz, x, blocks)
    assert rms(y[0::2], ref[0::2]) <= RMS_TOL
    # one NaN and one infinity in channel 1: that channel's block and the next (its overlap) are NaN like the reference's,
    # channel 0 does not notice
    g = ra.ResamplerFft.new(2, sr(in_hz), sr(out_hz))
    x = synth.fast_noise(blocks * n_in, seed=6).copy()
    x[1 + 2 * (3 * (n_in // 2) + 17)] = np.nan
    x[1 + 2 * (8 * (n_in // 2) + 5)] = np.inf
    with np.errstate(invalid="ignore", over="ignore"):
        ref = _oracle_blocks(in_hz, out_hz, x, blocks)
    y = g.resample_bulk(x, blocks)
    assert rms(y[0::2], ref[0::2]) <= RMS_TOL
    bad = np.zeros(blocks, bool)
    bad[[3, 4, 8, 9]] = True
    for b in range(blocks):
        yb, rb = y[b * n_out + 1:(b + 1) * n_out:2], ref[b * n_out + 1:(b + 1) * n_out:2]
        if bad[b]:
            assert not np.any(np.isfinite(rb)) and not np.any(np.isfinite(yb)), b
        else:
            assert np.all(np.isfinite(yb)) and rms(yb, rb) <= RMS_TOL, b


@pytest.mark.gpu
@pytest.mark.parametrize("in_hz,out_hz", [(44100, 48000), (48000, 44100)])
def test_a_batch_of_two_channel_streams_of_very_different_lengths(in_hz, out_hz):
    """One launch, streams of 1 .. 150 blocks: the two-channel kernel cuts every stream into pairs of a long and a short run
    sized for the LONGEST stream, so a short stream is a single (partial) long run, and most waves find nothing to do."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    lengths = [1, 7, 40, 3, 150, 2, 61]
    gs = [ra.ResamplerFft.new(2, sr(in_hz), sr(out_hz)) for _ in lengths]
    n_in, n_out = gs[0].chunk_size_input(), gs[0].chunk_size_output()
    xs = [synth.fast_noise(n * n_in, seed=10 + i) for i, n in enumerate(lengths)]
    d_in = [torch.from_numpy(x).to(dev) for x in xs]
    d_out = [torch.zeros(n * n_out, device=dev) for n in lengths]
    batch = ra.FftBatch(gs)
    for _round in range(2):   # (the second launch starts from the overlap the first left)
        batch.bind(d_in, d_out, lengths)
        batch.resample_bulk_device(ra.torch_stream())
        torch.cuda.synchronize()
        if _round == 0:
            first = [d.cpu().numpy().copy() for d in d_out]
    for i, n in enumerate(lengths):
        ref = _oracle_blocks(in_hz, out_hz, np.concatenate([xs[i], xs[i]]), 2 * n)
        assert rms(first[i], ref[:n * n_out]) <= RMS_TOL, i
        assert rms(d_out[i].cpu().numpy(), ref[n * n_out:]) <= RMS_TOL, i


@pytest.mark.gpu
def test_c3_full_size_and_batch_device_api():
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    # BASELINE config 3: 2 ch 44100 -> 48000, 892 blocks of 1176 frames (~2^20 frames).
    blocks = 892
    n_streams = 3
    gs = [ra.ResamplerFft.new(2, ra.SampleRate.Hz44100, ra.SampleRate.Hz48000) for _ in range(n_streams)]
    n_in, n_out = gs[0].chunk_size_input(), gs[0].chunk_size_output()
    xs = [synth.sweep(blocks * n_in // 2, 2, 44100.0) * np.float32(1.0 - 0.1 * i) for i in range(n_streams)]
    d_in = [torch.from_numpy(x).to(dev) for x in xs]
    d_out = [torch.zeros(blocks * n_out, device=dev) for _ in range(n_streams)]
    batch = ra.FftBatch(gs)
    batch.bind(d_in, d_out, [blocks] * n_streams)
    batch.resample_bulk_device(ra.torch_stream())
    torch.cuda.synchronize()
    for i in range(n_streams):
        r = o.OracleFft(2, 44100, 48000)
        ref = np.zeros((blocks, n_out), np.float32)
        for b in range(blocks):
            assert r.resample(xs[i][b * n_in:(b + 1) * n_in], ref[b]) == 0
        assert rms(d_out[i].cpu().numpy(), ref.reshape(-1)) <= RMS_TOL


@pytest.mark.gpu
def test_reference_amplitude_properties_on_gpu():
    # resampler_fft.rs:439-566 run against the GPU path itself.
    for in_hz, out_hz in [(48000, 44100), (44100, 48000), (48000, 32000), (32000, 48000),
                          (96000, 48000), (48000, 96000)]:
        g = ra.ResamplerFft.new(1, sr(in_hz), sr(out_hz))
        x = np.full(g.chunk_size_input(), 0.5, np.float32)
        out = np.zeros(g.chunk_size_output(), np.float32)
        for _ in range(5):
            g.resample(x, out)
        start = min(g.delay(), out.size // 4)
        assert np.abs(out[start:out.size * 3 // 4] - 0.5).max() < 0.02
    g = ra.ResamplerFft.new(2, ra.SampleRate.Hz48000, ra.SampleRate.Hz44100)
    x = np.zeros(g.chunk_size_input(), np.float32)
    x[0::2], x[1::2] = 0.3, 0.6
    out = np.zeros(g.chunk_size_output(), np.float32)
    for _ in range(5):
        g.resample(x, out)
    start = min(g.delay(), out.size // 8) * 2
    end = out.size * 3 // 4
    assert np.abs(out[start:end:2] - 0.3).max() < 0.02 and np.abs(out[start + 1:end:2] - 0.6).max() < 0.02


@pytest.mark.gpu
def test_whole_file_driver_matches_reference_loop():
    # resample_batch (resample/src/main.rs:256-313) -- zero-padded tail chunk and
    # ceil(len * out / in) trim -- replayed on the oracle chunk by chunk.
    g = ra.ResamplerFft.new(2, ra.SampleRate.Hz44100, ra.SampleRate.Hz48000)
    r = o.OracleFft(2, 44100, 48000)
    n_in, n_out = g.chunk_size_input(), g.chunk_size_output()
    x = synth.sweep((7 * n_in + 1000) // 2, 2, 44100.0)
    y = g.resample_batch(x)
    total = -(-x.size // n_in)
    padded = np.zeros(total * n_in, np.float32)
    padded[:x.size] = x
    ref = np.zeros((total, n_out), np.float32)
    for b in range(total):
        assert r.resample(padded[b * n_in:(b + 1) * n_in], ref[b]) == 0
    expected = int(np.ceil(x.size * n_out / n_in))
    assert y.size == expected
    assert rms(y, ref.reshape(-1)[:expected]) <= RMS_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("ch,in_hz,out_hz,blocks", [
    (4, 44100, 48000, 29), (8, 48000, 44100, 23), (6, 44100, 48000, 19),   # the 1176 <-> 1280 plans
    (4, 48000, 96000, 17), (8, 96000, 48000, 17),                           # 512-frame families
    (4, 22050, 48000, 13), (4, 88200, 96000, 9), (6, 44100, 192000, 7),     # long plans (one workgroup of a few waves)
    (16, 44100, 48000, 9),
])
def test_even_channel_counts_run_as_channel_pairs(ch, in_hz, out_hz, blocks):
    """Streams of 4, 6, 8 .. channels: channel pairs on the two-channel wave kernel (8-byte accesses at the frame's
    stride, the pair's waves exchange their final values through LDS).  Expectation: the reference run per channel;
    two bulk calls, so the second starts from a carried overlap and both contain more than one run of blocks."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    g = ra.ResamplerFft.new(ch, sr(in_hz), sr(out_hz))
    per_channel = [o.OracleFft(1, in_hz, out_hz) for _ in range(ch)]
    n_in, n_out = g.chunk_size_input(), g.chunk_size_output()
    x = synth.fast_noise(blocks * n_in, seed=ch + blocks)
    ref = np.zeros((blocks, n_out // ch, ch), np.float32)
    row = np.zeros(n_out // ch, np.float32)
    for k in range(blocks):
        xk = x[k * n_in:(k + 1) * n_in].reshape(-1, ch)
        for c in range(ch):
            assert per_channel[c].resample(np.ascontiguousarray(xk[:, c]), row) == 0
            ref[k, :, c] = row
    first = blocks // 3
    d_in = torch.from_numpy(x).to(dev)
    d_out = torch.zeros(blocks * n_out, device=dev)
    torch.cuda.synchronize()
    g.resample_bulk_device(d_in[:first * n_in], d_out[:first * n_out], first)
    g.resample_bulk_device(d_in[first * n_in:], d_out[first * n_out:], blocks - first)
    torch.cuda.synchronize()
    y = d_out.cpu().numpy()
    assert rms(y, ref.reshape(-1)) <= RMS_TOL, rms(y, ref.reshape(-1))
    assert float(np.max(np.abs(y - ref.reshape(-1)))) < 2e-5


@pytest.mark.gpu
def test_exact_build_is_bit_identical_to_the_reference_arithmetic():
    """libresampler_amd_fftexact.so = the same library with the wave kernel compiled without fused
    multiply-adds and with every twiddle fetched (make -C resampler_amd/csrc): its output equals the CPU
    path's bit for bit, both directions of the 44.1 / 48 kHz pair.  (One library per process: a child runs it.)"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exact = os.path.join(root, "resampler_amd", "libresampler_amd_fftexact.so")
    assert os.path.exists(exact), "build it: make -C resampler_amd/csrc"
    code = r"""
import numpy as np, torch
import resampler_amd as ra
from oracle import pyoracle as o
from resampler_amd import synth
assert ra.LIB_PATH.endswith("libresampler_amd_fftexact.so")
blocks = 40
for in_hz, out_hz, a, b in ((44100, 48000, ra.SampleRate.Hz44100, ra.SampleRate.Hz48000),
                            (48000, 44100, ra.SampleRate.Hz48000, ra.SampleRate.Hz44100)):
    g = ra.ResamplerFft.new(2, a, b)
    n_in, n_out = g.chunk_size_input(), g.chunk_size_output()
    x = synth.sweep(blocks * n_in // 2, 2, float(in_hz))
    dev = torch.device("cuda:0")
    d_in, d_out = [torch.from_numpy(x).to(dev)], [torch.zeros(blocks * n_out, device=dev)]
    batch = ra.FftBatch([g])
    batch.bind(d_in, d_out, [blocks])
    batch.resample_bulk_device(ra.torch_stream())
    torch.cuda.synchronize()
    r = o.OracleFft(2, in_hz, out_hz)
    ref = np.zeros((blocks, n_out), np.float32)
    for k in range(blocks):
        assert r.resample(x[k * n_in:(k + 1) * n_in], ref[k]) == 0
    y = d_out[0].cpu().numpy()
    assert np.array_equal(y, ref.reshape(-1)), (in_hz, float(np.abs(y - ref.reshape(-1)).max()))
# four channels (two channel pairs on the same code): the reference run per channel
g = ra.ResamplerFft.new(4, ra.SampleRate.Hz44100, ra.SampleRate.Hz48000)
n_in, n_out = g.chunk_size_input(), g.chunk_size_output()
x = synth.sweep(blocks * n_in // 4, 4, 44100.0)
d_in, d_out = torch.from_numpy(x).to(dev), torch.zeros(blocks * n_out, device=dev)
torch.cuda.synchronize()
g.resample_bulk_device(d_in, d_out, blocks)
torch.cuda.synchronize()
ref = np.zeros((blocks, n_out // 4, 4), np.float32)
row = np.zeros(n_out // 4, np.float32)
for c in range(4):
    r = o.OracleFft(1, 44100, 48000)
    for k in range(blocks):
        assert r.resample(np.ascontiguousarray(x[k * n_in:(k + 1) * n_in].reshape(-1, 4)[:, c]), row) == 0
        ref[k, :, c] = row
assert np.array_equal(d_out.cpu().numpy(), ref.reshape(-1)), "4 channels"
print("bit-identical")
"""
    env = dict(os.environ, RSMP_AMD_LIB=exact, PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "bit-identical" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("in_hz,out_hz,blocks,world", [(44100, 48000, 101, 4), (48000, 44100, 37, 3)])
def test_block_sharded_stream_equals_the_single_launch(in_hz, out_hz, blocks, world):
    """block ranges of one FFT stream on fresh resamplers, each recomputing one block of halo
    (sharding.run_fft_block_shard), equal the single bulk launch bit for bit."""
    torch = pytest.importorskip("torch")
    from resampler_amd import sharding
    dev = torch.device("cuda:0")
    mk = lambda: ra.ResamplerFft.new(2, sr(in_hz), sr(out_hz))
    g = mk()
    n_in, n_out = g.chunk_size_input(), g.chunk_size_output()
    x = synth.sweep(blocks * n_in // 2, 2, float(in_hz))
    d_x = torch.from_numpy(x).to(dev)
    d_whole = torch.zeros(blocks * n_out, device=dev)
    torch.cuda.synchronize()   # (launches run on the handles' own streams: order them after the fills)
    g.resample_bulk_device(d_x, d_whole, blocks)
    for first, end in sharding.fft_block_shards(blocks, world):
        d_work = torch.zeros((end - first + 1) * n_out, device=dev)
        torch.cuda.synchronize()
        d_out = sharding.run_fft_block_shard(mk(), first, end, d_x, d_work)
        torch.cuda.synchronize()
        assert d_out.numel() == (end - first) * n_out
        assert torch.equal(d_out, d_whole[first * n_out:end * n_out]), (first, end)


@pytest.mark.gpu
def test_every_rate_pair_of_the_reference_constructs_and_matches():
    """All 90 ordered pairs of the reference's SampleRate (src/lib.rs:167-188) through the bulk entry: one channel x
    three blocks (the any-channel-count builds, no halo) and two channels x 23 blocks (the two-channel builds; more
    than one run of blocks, so the halo recompute runs): blocks from 64 to 12288 frames, every kernel flavour
    (wave per transform, workgroup, one-buffer).  The two-channel expectation is the reference run per channel
    (channels = 1): with two channels its output scratch regions collide for 59 of the 90 pairs (`consistent`,
    SURVEY 7.3 item 6); where they do not, the two-channel reference is checked as well."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    worst = 0.0
    n_two_channel_reference = 0
    for a, b in itertools.permutations(RATES, 2):
        for ch, blocks in ((1, 3), (2, 23)):
            g = ra.ResamplerFft.new(ch, sr(a), sr(b))
            per_channel = [o.OracleFft(1, a, b) for _ in range(ch)]
            n_in, n_out = g.chunk_size_input(), g.chunk_size_output()
            assert (n_in, n_out, g.delay()) == (ch * per_channel[0].chunk_size_input(), ch * per_channel[0].chunk_size_output(),
                                                per_channel[0].delay()), (a, b)
            x = synth.fast_noise(blocks * n_in, seed=a % 977 + b % 13 + ch)
            d_out = torch.zeros(blocks * n_out, device=dev)
            d_in = torch.from_numpy(x).to(dev)
            torch.cuda.synchronize()   # (the launch runs on the handle's own stream: order it after the fills)
            g.resample_bulk_device(d_in, d_out, blocks)
            torch.cuda.synchronize()
            ref = np.zeros((blocks, n_out // ch, ch), np.float32)
            row = np.zeros(n_out // ch, np.float32)
            for k in range(blocks):
                xk = x[k * n_in:(k + 1) * n_in].reshape(-1, ch)
                for c in range(ch):
                    assert per_channel[c].resample(np.ascontiguousarray(xk[:, c]), row) == 0
                    ref[k, :, c] = row
            y = d_out.cpu().numpy()
            e = rms(y, ref.reshape(-1))
            assert e <= RMS_TOL, (a, b, ch, e)
            worst = max(worst, e)
            if ch > 1 and consistent(ch, a, b):
                r = o.OracleFft(ch, a, b)
                whole = np.zeros((blocks, n_out), np.float32)
                for k in range(blocks):
                    assert r.resample(x[k * n_in:(k + 1) * n_in], whole[k]) == 0
                assert np.array_equal(whole.reshape(-1), ref.reshape(-1)), (a, b)
                n_two_channel_reference += 1
    assert worst > 0.0 and n_two_channel_reference == 31
