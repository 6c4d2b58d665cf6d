"""GPU parity tests of the FIR path: everything goes through the C ABI (libresampler_amd.so) and is
compared with the CPU oracle on the same inputs.  Gate (BASELINE.json north_star): output within
1e-6 RMS of the CPU path, identical (consumed, produced) counts."""
import os

import numpy as np
import pytest

import resampler_amd as ra
from oracle import pyoracle as o
from resampler_amd import synth

pytestmark = pytest.mark.gpu

RMS_TOL = 1e-6   # north_star tolerance
def _knob(name, default):
    """An A/B switch of the library as the library sees it: only under RSMP_DEBUG=1 (csrc/common.h, rsmp::knob)."""
    return os.environ.get(name, default) if os.environ.get("RSMP_DEBUG", "0") not in ("", "0") else default


SPLIT_VARIANT = 4 if _knob("RSMP_FIR_SPLIT_PLANES", "2") == "3" else 5   # bf16x3 or (default) fp16x2 split kernel
ATT_DB = {ra.Attenuation.Db60: 60, ra.Attenuation.Db90: 90, ra.Attenuation.Db120: 120}


def rms(a, b):
    if a.size == 0:
        return 0.0
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


ORACLE_KIND = o.CONVOLVE_AVX_FMA if o.have_avx_fma() else o.CONVOLVE_SCALAR   # north_star: "matches the reference CPU SIMD path"


def make_pair(ch, in_hz, out_hz, lat=ra.Latency.Sample64, att=ra.Attenuation.Db90, kernel=None):
    """The HIP resampler and the oracle on convolve_interp_avx_fma (src/fir/avx.rs:5-61), the path north_star names
    (scalar spec, src/fir/mod.rs:47-62, on a CPU without AVX + FMA)."""
    g = ra.ResamplerFir.new_from_hz(ch, in_hz, out_hz, lat, att)
    if kernel is not None:
        g.set_kernel(kernel)
    r = o.OracleFir(ch, in_hz, out_hz, lat.taps(), ATT_DB[att], ORACLE_KIND)
    return g, r


def stream_compare(g, r, x, chunks, out_caps=None):
    """Feeds the same slices to both and compares every call."""
    ch = g.channels
    og = np.zeros(g.buffer_size_output(), np.float32)
    orr = np.zeros(r.buffer_size_output(), np.float32)
    off = 0
    worst = 0.0
    i = 0
    while off < x.size:
        n = chunks[i % len(chunks)] * ch
        cap = og.size if out_caps is None else min(og.size, out_caps[i % len(out_caps)] * ch)
        sl = x[off:off + n]
        cg, pg = g.resample(sl, og[:cap])
        rc, cr, pr = r.resample(sl, orr[:cap])
        assert rc == 0
        assert (cg, pg) == (cr, pr), (i, (cg, pg), (cr, pr))
        e = rms(og[:pg], orr[:pr])
        if e > RMS_TOL:   # diagnostic for the failure report
            bad = np.flatnonzero(np.abs(og[:pg].astype(np.float64) - orr[:pr]) > 1e-4)
            print(f"stream_compare: call {i} consumed {cg} produced {pg} rms {e:.3e} bad values {bad.size}: "
                  f"{bad[:8]} ... {bad[-4:]} got {og[bad[:4]]} want {orr[bad[:4]]}")
        worst = max(worst, e)
        off += cg
        i += 1
        if cg == 0 and pg == 0:
            break
    return worst


def test_c1_plumbing_config_on_gpu():
    # BASELINE config 1 shape: 1 ch 48000 -> 44100, Sample64 / Db90, 512-sample chunks.
    g, r = make_pair(1, 48000, 44100)
    x = synth.sweep(1 << 20, 1, 48000.0)   # 2^20 frames
    assert g.buffer_size_output() == r.buffer_size_output() == 3648
    assert g.delay() == r.delay() == 64
    assert stream_compare(g, r, x, [512]) <= RMS_TOL


@pytest.mark.parametrize("ch,in_hz,out_hz,lat,att", [
    (2, 44100, 48000, ra.Latency.Sample64, ra.Attenuation.Db90),
    (2, 48000, 44100, ra.Latency.Sample64, ra.Attenuation.Db120),
    (8, 96000, 44100, ra.Latency.Sample64, ra.Attenuation.Db120),
    (3, 44100, 96000, ra.Latency.Sample32, ra.Attenuation.Db60),
    (1, 24000, 16000, ra.Latency.Sample16, ra.Attenuation.Db60),
    (2, 44100, 48001, ra.Latency.Sample8, ra.Attenuation.Db90),
    (5, 384000, 16000, ra.Latency.Sample64, ra.Attenuation.Db90),
])
def test_streaming_calls_match_oracle(ch, in_hz, out_hz, lat, att):
    g, r = make_pair(ch, in_hz, out_hz, lat, att)
    x = synth.fast_noise(ch * 30000, seed=ch)
    # ragged chunk sizes incl. empty and > INPUT_CAPACITY, and small output buffers
    worst = stream_compare(g, r, x, [256, 1, 0, 4096, 5000, 17, 512],
                           out_caps=[100000, 100000, 64, 100000, 7, 100000])
    assert worst <= RMS_TOL
    # reset(): same behaviour afterwards
    g.reset()
    r.reset()
    assert stream_compare(g, r, x[: ch * 5000], [333]) <= RMS_TOL


def test_error_codes_match_reference():
    g, r = make_pair(3, 44100, 48000)
    out = np.zeros(g.buffer_size_output(), np.float32)
    with pytest.raises(ra.InvalidInputBufferSize):
        g.resample(np.zeros(100, np.float32), out)
    with pytest.raises(ra.InvalidOutputBufferSize):
        g.resample(np.zeros(99, np.float32), out[:100])
    # input is validated first (resampler_fir.rs:514-519)
    with pytest.raises(ra.InvalidInputBufferSize):
        g.resample(np.zeros(100, np.float32), out[:100])
    assert g.resample(np.zeros(0, np.float32), out[:0]) == (0, 0)


@pytest.mark.parametrize("kernel", [ra.FirKernel.Generic, ra.FirKernel.Periodic, ra.FirKernel.PeriodicVector,
                                    ra.FirKernel.PeriodicF32])
@pytest.mark.parametrize("ch,in_hz,out_hz,att", [
    (2, 44100, 48000, ra.Attenuation.Db90),
    (2, 48000, 44100, ra.Attenuation.Db90),
    (1, 44100, 96000, ra.Attenuation.Db120),
    (2, 96000, 44100, ra.Attenuation.Db120),
    (2, 48000, 96000, ra.Attenuation.Db90),
    (2, 96000, 48000, ra.Attenuation.Db90),
    (8, 96000, 44100, ra.Attenuation.Db120),
    (2, 44100, 96000, ra.Attenuation.Db90),
    (4, 44100, 96000, ra.Attenuation.Db90),
    (2, 88200, 44100, ra.Attenuation.Db90),
    (2, 192000, 48000, ra.Attenuation.Db90),
    (8, 48000, 96000, ra.Attenuation.Db90),    # channel pairs of an exact ratio: no wrap windows fetched
    (6, 96000, 48000, ra.Attenuation.Db90),    # ... two rounds, windows by take (none)
    (4, 22050, 48000, ra.Attenuation.Db60),
    # more than two channels at 147/160: even counts up to 16 run the split matrix kernel as channel pairs, the
    # others the vector kernel
    (4, 44100, 48000, ra.Attenuation.Db90),
    (6, 48000, 44100, ra.Attenuation.Db90),
    (8, 44100, 48000, ra.Attenuation.Db120),
    (3, 44100, 48000, ra.Attenuation.Db90),
    (12, 48000, 44100, ra.Attenuation.Db90),
    (16, 44100, 48000, ra.Attenuation.Db90),
    (1, 48000, 44100, ra.Attenuation.Db90),     # BASELINE config 1's stream (vector kernel)
])
def test_bulk_matches_reference_driver_loop(kernel, ch, in_hz, out_hz, att):
    g, r = make_pair(ch, in_hz, out_hz, att=att, kernel=kernel)
    x = synth.sweep(60000, ch, float(in_hz))
    chunk = 512 - (512 % ch)
    yg, consumed, calls_g = g.resample_bulk(x, chunk, want_calls=True)
    yr, calls_r = r.resample_all(x, chunk)
    assert consumed == int(calls_r[:, 0].sum())
    assert np.array_equal(calls_g, calls_r)
    assert yg.size == yr.size
    assert rms(yg, yr) <= RMS_TOL
    # the stream continues seamlessly: per-call API after a bulk launch, then bulk again
    x2 = synth.fast_noise(ch * 9000, seed=5)
    assert stream_compare(g, r, x2[: ch * 3000], [700]) <= RMS_TOL
    yg2, _ = g.resample_bulk(x2[ch * 3000:], chunk)
    yr2, _ = r.resample_all(x2[ch * 3000:], chunk)
    assert yg2.size == yr2.size and rms(yg2, yr2) <= RMS_TOL
    if (kernel == ra.FirKernel.Periodic and ch in (1, 2, 3, 4, 6, 8, 12, 16) and {in_hz, out_hz} == {44100, 48000}
            and _knob("RSMP_FIR_MFMA", "3") == "3" and _knob("RSMP_FIR_SPLIT_WIDE", "1") != "0"):
        assert g.kernel_variant() == SPLIT_VARIANT
    # the other rate pairs of config 4 and config 5's: tile groups (up to 320 classes), two rounds of lane tasks (periods of
    # up to 320 frames), super periods of exact ratios (48 <-> 96 kHz)
    if (kernel == ra.FirKernel.Periodic and (ch, in_hz, out_hz) in ((2, 96000, 44100), (8, 96000, 44100), (2, 44100, 96000), (4, 44100, 96000), (2, 48000, 96000), (2, 96000, 48000), (2, 88200, 44100), (2, 192000, 48000), (8, 48000, 96000), (6, 96000, 48000))
            and _knob("RSMP_FIR_MFMA", "3") == "3" and _knob("RSMP_FIR_SPLIT_LONG", "1") != "0" and SPLIT_VARIANT == 5):
        assert g.kernel_variant() == SPLIT_VARIANT
    if kernel == ra.FirKernel.PeriodicVector:
        assert g.kernel_variant() in (1, 2)   # never the matrix-core kernel
    if kernel == ra.FirKernel.PeriodicF32:
        assert g.kernel_variant() in (1, 2, 3)   # never the split-bf16 kernel
    if kernel == ra.FirKernel.Generic:
        assert g.kernel_variant() == 0


@pytest.mark.parametrize("ch,in_hz,out_hz", [(4, 44100, 48000), (8, 48000, 44100), (6, 44100, 48000), (1, 48000, 44100), (3, 44100, 48000),
                                             (5, 48000, 44100)])
def test_split_kernel_channel_pairs_long_stream(ch, in_hz, out_hz):
    """4 / 8 channels at 147/160 on the split matrix kernel (an item = one channel pair of a block): a stream long
    enough that a workgroup gets both pairs of a block (16-byte loads shared by two items, the even pair's sums
    stored together with the odd pair's) as well as blocks cut between two workgroups; non-finite and out-of-range
    samples in single channels go through the repair launch."""
    g, _ = make_pair(ch, in_hz, out_hz, kernel=ra.FirKernel.Periodic)
    r = o.OracleFir(ch, in_hz, out_hz, 128, 90, o.CONVOLVE_AVX_FMA if o.have_avx_fma() else o.CONVOLVE_SCALAR)
    n = 700_000
    x = synth.fast_noise(ch * n, seed=31 + ch)
    x[ch * 1000 + min(2, ch - 1)] = np.inf
    x[ch * 300_000 + min(1, ch - 1)] = np.nan
    x[ch * 500_000 + min(3, ch - 1)] = 1000.0        # beyond the two-plane fp16 split
    chunk = 512 * ch
    yg, consumed = g.resample_bulk(x, chunk)
    yr, _ = r.resample_all(x, chunk)
    assert consumed == x.size and yg.size == yr.size
    if _knob("RSMP_FIR_MFMA", "3") == "3" and _knob("RSMP_FIR_SPLIT_WIDE", "1") != "0":
        assert g.kernel_variant() == SPLIT_VARIANT
    fin_r, fin_g = np.isfinite(yr), np.isfinite(yg)
    assert np.array_equal(fin_g, fin_r), np.flatnonzero(fin_g != fin_r)[:10]
    assert (~fin_r).any()
    ok = fin_r
    assert rms(yg[ok], yr[ok]) <= RMS_TOL * max(1.0, float(np.sqrt(np.mean(yr[ok].astype(np.float64) ** 2))))
    # every channel carries its own signal: per channel as well
    for c in range(ch):
        a, b = yg[c::ch], yr[c::ch]
        m = np.isfinite(b)
        assert rms(a[m], b[m]) <= RMS_TOL * max(1.0, float(np.sqrt(np.mean(b[m].astype(np.float64) ** 2)))), c


@pytest.mark.parametrize("taps_lat", [ra.Latency.Sample8, ra.Latency.Sample16, ra.Latency.Sample32])
def test_bulk_periodic_other_tap_counts(taps_lat):
    g, r = make_pair(2, 44100, 48000, lat=taps_lat, kernel=ra.FirKernel.Periodic)
    x = synth.sweep(50000, 2, 44100.0)
    yg, _ = g.resample_bulk(x, 512)
    yr, _ = r.resample_all(x, 512)
    assert yg.size == yr.size and rms(yg, yr) <= RMS_TOL


def test_c2_full_size_bulk_parity_and_max_abs():
    # BASELINE config 2: 2 ch interleaved 44100 -> 48000, 128 taps, 2^20-frame sine sweep, against the oracle's
    # convolve_interp_avx_fma (the CPU SIMD path north_star names; make_pair).
    g, r = make_pair(2, 44100, 48000)
    x = synth.sweep(1 << 20, 2, 44100.0)
    yg, consumed, calls_g = g.resample_bulk(x, 512, want_calls=True)
    yr, calls_r = r.resample_all(x, 512)
    assert consumed == x.size
    assert np.array_equal(calls_g, calls_r)
    assert yg.size == yr.size
    assert rms(yg, yr) <= RMS_TOL
    assert float(np.max(np.abs(yg.astype(np.float64) - yr))) < 2e-5
    if _knob("RSMP_FIR_MFMA", "3") == "3":
        assert g.kernel_variant() == SPLIT_VARIANT   # the full-size config runs on the split matrix kernel (the bench's kernel)


def rel_rms(a, b):
    return rms(a, b) / max(float(np.sqrt(np.mean(b.astype(np.float64) ** 2))), 1e-300)


@pytest.mark.parametrize("ch,in_hz,out_hz", [(2, 44100, 48000), (2, 48000, 44100), (4, 44100, 48000), (8, 48000, 44100),
                                             (1, 48000, 44100), (3, 44100, 48000), (8, 96000, 44100), (2, 44100, 96000),
                                             (2, 48000, 96000), (2, 96000, 48000)])
@pytest.mark.parametrize("level", ["2^-10", "2^-17", "2^-24", "1e-30", "3e4", "mix"])
def test_quiet_and_loud_signals_keep_their_relative_precision(ch, in_hz, out_hz, level):
    """The gate is 1e-6 RMS on full-scale audio; a drop-in for an f32 resampler must hold it RELATIVE to the signal at
    any level: the C2 sweep scaled by 2^-10, 2^-17, 2^-24 (one f32 ulp of full scale), 1e-30, 3e4 (far outside
    [-1, 1]), and a full-scale sweep carrying a -100 dBFS one ("mix": after subtracting the loud part's own output the
    quiet part must still be there to 1e-3 of ITS level).  The split matrix kernels cut every item's samples into two
    fp16 planes as block floating point (fir_split.hip, `peak`), so all of these keep f32-class precision."""
    g, r = make_pair(ch, in_hz, out_hz, kernel=ra.FirKernel.Periodic)
    n = 70000
    x = synth.sweep(n, ch, float(in_hz))
    if level == "mix":
        q = synth.fast_noise(n * ch, seed=11) * np.float32(1e-5)
        xm = (x + q).astype(np.float32)
        yg, _ = g.resample_bulk(xm, 512 - 512 % ch)
        yr, _ = r.resample_all(xm, 512 - 512 % ch)
        assert yg.size == yr.size and rel_rms(yg, yr) <= RMS_TOL
        # the quiet component alone: the oracle's response to (x + q) minus its response to x
        r2 = o.OracleFir(ch, in_hz, out_hz, 128, 90, ORACLE_KIND)
        y0, _ = r2.resample_all(x, 512 - 512 % ch)
        quiet_ref = yr.astype(np.float64) - y0
        quiet_got = yg.astype(np.float64) - y0
        # (what is left is the f32 rounding of the loud part, ~1e-7 of ITS level)
        assert rms(quiet_got, quiet_ref) <= max(2e-2 * float(np.sqrt(np.mean(quiet_ref ** 2))), 3e-7 * float(np.sqrt(np.mean(yr.astype(np.float64) ** 2))))
        return
    scale = {"2^-10": 2.0 ** -10, "2^-17": 2.0 ** -17, "2^-24": 2.0 ** -24, "1e-30": 1e-30, "3e4": 3e4}[level]
    xs = (x * np.float32(scale)).astype(np.float32)
    yg, consumed = g.resample_bulk(xs, 512 - 512 % ch)
    yr, _ = r.resample_all(xs, 512 - 512 % ch)
    assert consumed == xs.size and yg.size == yr.size
    assert rel_rms(yg, yr) <= RMS_TOL, (level, rel_rms(yg, yr))


def test_device_resident_api_and_batch():
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    n_streams = 7
    pairs = [(44100, 48000), (48000, 44100), (44100, 96000)]
    gs, rs, xs = [], [], []
    for i in range(n_streams):
        in_hz, out_hz = pairs[i % len(pairs)]
        g, r = make_pair(2, in_hz, out_hz)
        gs.append(g)
        rs.append(r)
        xs.append(synth.fast_noise(2 * (30000 + 1000 * (i % 2)), seed=100 + i))
    batch = ra.FirBatch(gs)
    for step in range(2):
        d_in = [torch.from_numpy(x).to(dev) for x in xs]
        d_out = [torch.zeros(g.bulk_output_bound(x.size, 512), device=dev) for g, x in zip(gs, xs)]
        batch.bind(d_in, d_out)
        consumed, produced = batch.resample_bulk_device(512, ra.torch_stream())
        torch.cuda.synchronize()
        for i in range(n_streams):
            yr, calls = rs[i].resample_all(xs[i], 512)
            assert consumed[i] == xs[i].size and produced[i] == yr.size
            assert rms(d_out[i][: produced[i]].cpu().numpy(), yr) <= RMS_TOL
    # single-stream device call
    g, r = make_pair(2, 44100, 48000)
    x = synth.fast_noise(2 * 2000, seed=9)
    d_x = torch.from_numpy(x).to(dev)
    d_y = torch.zeros(g.buffer_size_output(), device=dev)
    c, p = g.resample_device(d_x, d_y, ra.torch_stream())
    torch.cuda.synchronize()
    orr = np.zeros(r.buffer_size_output(), np.float32)
    rc, cr, pr = r.resample(x, orr)
    assert (c, p) == (cr, pr) and rms(d_y[:p].cpu().numpy(), orr[:pr]) <= RMS_TOL


@pytest.mark.parametrize("kernel", [ra.FirKernel.Periodic, ra.FirKernel.PeriodicF32])
def test_matrix_core_kernel_ragged_batch_and_edges(kernel):
    """The default kernels for 2-channel rational rate pairs are the matrix-core periodic kernels
    (split-bf16 for 44.1 <-> 48 kHz, exact f32 when asked for or elsewhere): check that
    the right one runs, on a ragged batch whose streams are
    shorter than a period, one frame past a period block, and long; then a second launch on the
    carried state (wrap bitmap, class-table drift and hist/in junction all in play)."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    frames = [1, 50, 147, 148, 9408, 9409, 20000, 64 * 147 * 3 + 77, 123457]
    gs, rs, xs = [], [], []
    for i, n in enumerate(frames):
        g, r = make_pair(2, 44100, 48000, kernel=kernel)
        gs.append(g)
        rs.append(r)
        xs.append(synth.fast_noise(2 * n, seed=300 + i))
    batch = ra.FirBatch(gs)
    for step in range(2):
        d_in = [torch.from_numpy(x).to(dev) for x in xs]
        d_out = [torch.zeros(g.bulk_output_bound(x.size, 512), device=dev) for g, x in zip(gs, xs)]
        batch.bind(d_in, d_out)
        consumed, produced = batch.resample_bulk_device(512, ra.torch_stream())
        torch.cuda.synchronize()
        for i in range(len(frames)):
            yr, _ = rs[i].resample_all(xs[i], 512)
            assert consumed[i] == xs[i].size and produced[i] == yr.size, (step, i)
            assert rms(d_out[i][: produced[i]].cpu().numpy(), yr) <= RMS_TOL, (step, i)
    knob = _knob("RSMP_FIR_MFMA", "3")
    mfma_on = knob != "0"
    # periodic matrix-core kernel by default (4 = the split-bf16 one, RSMP_FIR_MFMA=3)
    split = knob == "3" and kernel == ra.FirKernel.Periodic
    assert gs[-1].kernel_variant() == ((SPLIT_VARIANT if split else 3) if mfma_on else 1)
    # other even rate pairs on the same path: 44.1 -> 96 k (20 class tiles), 16 / 32 / 64 taps above
    g, r = make_pair(2, 44100, 96000, kernel=kernel)
    x = synth.sweep(70001, 2, 44100.0)
    yg, _ = g.resample_bulk(x, 512)
    yr, _ = r.resample_all(x, 512)
    assert yg.size == yr.size and rms(yg, yr) <= RMS_TOL
    long_split = split and _knob("RSMP_FIR_SPLIT_LONG", "1") != "0" and SPLIT_VARIANT == 5   # (two tile groups on the split kernel)
    assert g.kernel_variant() == ((SPLIT_VARIANT if long_split else 3) if mfma_on else 1)


def test_batch_with_two_rate_pairs_and_mixed_kernels():
    """One batch launch holding streams of both split-kernel rate pairs (44.1 -> 48 k and 48 -> 44.1 k: two
    kernel groups, so the tails are copied by the separate launch) plus a 44.1 -> 96 k stream (f32 matrix-core
    kernel) and a mono stream (vector kernel); two launches so that the carried state is exercised too."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    spec = [(2, 44100, 48000, 30011), (2, 48000, 44100, 41000), (2, 44100, 48000, 5000),
            (2, 44100, 96000, 20000), (1, 48000, 44100, 33000), (2, 48000, 44100, 160 * 16 * 3)]
    gs, rs, xs = [], [], []
    for i, (ch, a, b, n) in enumerate(spec):
        g, r = make_pair(ch, a, b, kernel=ra.FirKernel.Periodic)
        gs.append(g)
        rs.append(r)
        xs.append(synth.fast_noise(ch * n, seed=900 + i))
    batch = ra.FirBatch(gs)
    for step in range(2):
        d_in = [torch.from_numpy(x).to(dev) for x in xs]
        d_out = [torch.zeros(g.bulk_output_bound(x.size, 512), device=dev) for g, x in zip(gs, xs)]
        batch.bind(d_in, d_out)
        consumed, produced = batch.resample_bulk_device(512, ra.torch_stream())
        torch.cuda.synchronize()
        for i in range(len(spec)):
            chunk = 512 - 512 % spec[i][0]
            yr, _ = rs[i].resample_all(xs[i], chunk)
            assert consumed[i] == xs[i].size and produced[i] == yr.size, (step, i)
            assert rms(d_out[i][: produced[i]].cpu().numpy(), yr) <= RMS_TOL, (step, i)


def test_repeated_launches_are_bit_identical():
    """The periodic kernels are full of dynamic scheduling (work queues, claims, producer / consumer
    flags): whatever the interleaving, a launch must produce the same bits.  40 launches of a
    16-stream batch, each compared with the first one (which is checked against the oracle)."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    n_streams, frames = 16, 150001
    gs, xs = [], []
    for i in range(n_streams):
        gs.append(ra.ResamplerFir.new_from_hz(2, 44100, 48000, ra.Latency.Sample64, ra.Attenuation.Db90))
        xs.append(synth.fast_noise(2 * frames, seed=700 + i))
    d_in = [torch.from_numpy(x).to(dev) for x in xs]
    d_out = [torch.zeros(gs[0].bulk_output_bound(2 * frames, 512), device=dev) for _ in range(n_streams)]
    batch = ra.FirBatch(gs)
    batch.bind(d_in, d_out)
    stream = ra.torch_stream()
    first = None
    for launch in range(40):
        for o_ in d_out:
            o_.fill_(float("nan"))
        batch.reset()
        consumed, produced = batch.resample_bulk_device(512, stream)
        torch.cuda.synchronize()
        got = [d_out[i][: produced[i]].clone() for i in range(n_streams)]
        if first is None:
            first = got
            r = o.OracleFir(2, 44100, 48000, 128, 90)
            yr, _ = r.resample_all(xs[0], 512)
            assert produced[0] == yr.size and rms(first[0].cpu().numpy(), yr) <= RMS_TOL
        else:
            for i in range(n_streams):
                assert torch.equal(got[i], first[i]), (launch, i)


def test_fuzz_periodic_kernels_against_oracle():
    """Seeded sweep over audio rate pairs, tap counts, channel counts, lengths and chunk sizes on the
    periodic kernels (matrix-core for 2 channels where two images fit, vector otherwise): counts
    identical, output within the gate, and a second launch continuing each stream."""
    rates = [8000, 11025, 16000, 22050, 32000, 44100, 48000, 88200, 96000, 176400, 192000]
    lats = [ra.Latency.Sample8, ra.Latency.Sample16, ra.Latency.Sample32, ra.Latency.Sample64]
    rng = np.random.default_rng(20240607)
    variants = set()
    for case in range(28):
        in_hz, out_hz = (int(v) for v in rng.choice(rates, 2, replace=False))
        ch = int(rng.choice([1, 2, 2, 2, 3, 4]))
        lat = lats[int(rng.integers(len(lats)))]
        g, r = make_pair(ch, in_hz, out_hz, lat=lat, kernel=ra.FirKernel.Periodic)
        chunk = int(rng.integers(1, 300)) * ch
        for part in range(2):
            frames = int(rng.integers(1, 40000))
            x = synth.fast_noise(ch * frames, seed=1000 + 10 * case + part)
            yg, consumed, calls_g = g.resample_bulk(x, chunk, want_calls=True)
            yr, calls_r = r.resample_all(x, chunk)
            assert np.array_equal(calls_g, calls_r), (case, part, in_hz, out_hz, ch)
            assert yg.size == yr.size, (case, part)
            assert rms(yg, yr) <= RMS_TOL, (case, part, in_hz, out_hz, ch, lat)
        variants.add(g.kernel_variant())
    assert 1 in variants                      # the vector kernel was exercised ...
    if _knob("RSMP_FIR_MFMA", "3") != "0":
        assert variants & {3, 4}              # ... and so were the matrix-core kernels (unless switched off)


def test_linearity_and_shift_properties_at_full_size():
    # Size-independent properties on a large launch: linearity, and identical channels in ->
    # identical channels out.
    g1 = ra.ResamplerFir.new(2, ra.SampleRate.Hz44100, ra.SampleRate.Hz48000, ra.Latency.Sample64,
                             ra.Attenuation.Db90)
    g2 = ra.ResamplerFir.new(2, ra.SampleRate.Hz44100, ra.SampleRate.Hz48000, ra.Latency.Sample64,
                             ra.Attenuation.Db90)
    g3 = ra.ResamplerFir.new(2, ra.SampleRate.Hz44100, ra.SampleRate.Hz48000, ra.Latency.Sample64,
                             ra.Attenuation.Db90)
    n = 1 << 19
    a = synth.fast_noise(2 * n, seed=1)
    b = synth.fast_noise(2 * n, seed=2)
    ya, _ = g1.resample_bulk(a, 512)
    yb, _ = g2.resample_bulk(b, 512)
    yab, _ = g3.resample_bulk((a + b).astype(np.float32), 512)
    assert ya.size == yb.size == yab.size
    assert rms(yab, ya + yb) <= 2e-6
    mono = synth.fast_noise(n, seed=3)
    st = np.repeat(mono, 2)
    g4 = ra.ResamplerFir.new(2, ra.SampleRate.Hz44100, ra.SampleRate.Hz48000)
    y, _ = g4.resample_bulk(st, 512)
    assert np.array_equal(y[0::2], y[1::2])


def test_long_stream_keeps_parity_while_the_f64_position_drifts():
    # 24 x 2^20 frames through one stream: the f64 position drifts away from the exact rational
    # position (~4e-9 per 2^20 frames), the periodic kernel's class table is rebuilt on the way and
    # the wrap bitmap changes character; counts and samples must keep matching the oracle.
    g, r = make_pair(2, 44100, 48000)
    r_avx = o.OracleFir(2, 44100, 48000, 128, 90, o.CONVOLVE_AVX_FMA) if o.have_avx_fma() else r
    x = synth.fast_noise(2 << 20, seed=11)
    worst = 0.0
    for i in range(24):
        yg, consumed, calls_g = g.resample_bulk(x, 512, want_calls=True)
        yr, calls_r = r_avx.resample_all(x, 512)
        assert consumed == x.size and np.array_equal(calls_g, calls_r), i
        assert yg.size == yr.size
        worst = max(worst, rms(yg, yr))
    assert worst <= RMS_TOL
    assert g.state() == r_avx.state()   # read_position, available_frames and the drifted f64 position, bit for bit


@pytest.mark.parametrize("in_hz,out_hz", [(22050, 44100), (22050, 48000)])
def test_stopband_attenuation_on_gpu(in_hz, out_hz):
    # the reference's own quality gate (resampler_fir.rs:693-815) run through the
    # GPU path: impulse in 256-value calls, >= 90 dB between pass band and stop band.
    n = int(np.float32(in_hz) * np.float32(5.0))
    x = np.zeros(n, np.float32)
    x[min(int(n * 0.5), n - 1)] = 1.0
    g = ra.ResamplerFir.new_from_hz(1, in_hz, out_hz, ra.Latency.Sample64, ra.Attenuation.Db90)
    y, consumed = g.resample_bulk(x, 256)
    assert consumed == n
    peak = int(np.argmax(np.abs(y)))
    win = int(np.float32(out_hz) * np.float32(0.1))
    start = max(0, peak - win // 2)
    seg = y[start:min(start + win, len(y))][:8192]
    mag_db = 20.0 * np.log10(np.maximum(np.abs(np.fft.rfft(seg.astype(np.float64), 8192)), 1e-10))

    def to_bin(f):
        return int(round(f * 8192 / out_hz))
    nyq = in_hz / 2.0
    pass_max = mag_db[to_bin(20.0):to_bin(nyq * 0.9) + 1].max()
    stop_max = mag_db[to_bin(nyq * 1.1):min(len(mag_db) - 10, to_bin(out_hz / 2.0 * 0.95)) + 1].max()
    assert pass_max - stop_max >= 90.0


@pytest.mark.parametrize("kernel", [ra.FirKernel.Generic, ra.FirKernel.Periodic, ra.FirKernel.PeriodicF32,
                                    ra.FirKernel.PeriodicVector])
def test_non_finite_and_huge_input_match_the_reference(kernel):
    """Inf / NaN samples (stream start, mid tile, tile edge, one channel and both) and samples far outside the
    audio range.  The periodic kernels pre-mix phase rows, pad windows with zero coefficients and (split
    kernel) cut operands into 16-bit planes; every store path therefore checks its sums and a repair launch
    re-evaluates the marked 1024-frame chunks in the reference's own form (fir_nonfinite.h).  Result: the same
    outputs are finite, +inf, -inf or NaN as in the reference (src/fir/avx.rs:25-58), on every kernel."""
    g, _ = make_pair(2, 44100, 48000, kernel=kernel)
    r = o.OracleFir(2, 44100, 48000, 128, 90, o.CONVOLVE_AVX_FMA if o.have_avx_fma() else o.CONVOLVE_SCALAR)
    n = 60000
    x = synth.fast_noise(2 * n, seed=77)
    x[0] = np.inf                      # first sample of the stream, channel 0
    x[2 * 5000 + 1] = -np.inf          # channel 1
    x[2 * 11760] = np.nan              # frame 11760 = 80 input periods of 147: a period edge
    x[2 * 11760 + 1] = np.nan
    x[2 * 30007] = np.inf
    x[2 * 30011] = -np.inf             # +inf and -inf inside one window
    x[2 * 47040 - 2] = np.inf          # last frame of a 16-period item (16 * 147 * 20 = 47040)
    x[2 * 20000] = 1000.0              # finite, but beyond the range of the two-plane fp16 split
    x[2 * 20001 + 1] = -3.0e30
    x[2 * 25000] = 15.99               # just inside it
    x[2 * 55000 + 1] = 1.0e-30         # and far below it
    yg, consumed = g.resample_bulk(x, 512)
    yr, _ = r.resample_all(x, 512)
    assert consumed == x.size and yg.size == yr.size
    fin_r, fin_g = np.isfinite(yr), np.isfinite(yg)
    assert np.array_equal(fin_g, fin_r), np.flatnonzero(fin_g != fin_r)[:10]
    # inf versus NaN: identical, except -- on the periodic kernels -- at outputs whose exact position is a
    # multiple of 1/1024 frame (every 5th output here): there the reference's frac is its f64 rounding
    # residue (~1e-12) times 1024, i.e. 0 or not by the sign of that residue, and inf * frac is NaN or inf
    # accordingly; the repair pass knows the position to the class table's 2e-9 drift quantum only.
    kind = np.isnan(yg) != np.isnan(yr)
    if kernel == ra.FirKernel.Generic:
        assert not kind.any()
    else:
        m = np.flatnonzero(kind) // 2
        assert np.all((m * 147 * 1024) % 160 == 0), m[:10]
    inf = np.isinf(yr) & np.isinf(yg)
    assert inf.any() and np.array_equal(yg[inf], yr[inf])          # same signed infinities
    # finite outputs: relative to the local magnitude (the 1e30 sample makes finite outputs of ~1e29)
    big = fin_r & (np.abs(yr) > 1e3)
    assert big.any() and np.all(np.abs(yg[big] - yr[big]) <= 1e-5 * np.abs(yr[big]))
    small = fin_r & ~big
    assert rms(yg[small], yr[small]) <= RMS_TOL


@pytest.mark.parametrize("in_hz,out_hz", [(96000, 44100), (192000, 44100), (384000, 16000)])
@pytest.mark.parametrize("kernel", [ra.FirKernel.Periodic, ra.FirKernel.PeriodicVector])
def test_non_finite_input_with_few_taps_and_heavy_downsampling(kernel, in_hz, out_hz):
    """Sample8 (16 taps) while down-sampling: one inf / NaN sample spoils only ~16 * out / in consecutive outputs
    (7, 4 and fewer than 1 here) -- fewer than the frames a lane of the periodic kernels stores, so the store
    paths test EVERY frame they write (fir_nonfinite.h), not one per lane.  Same finite / inf / NaN pattern as
    the reference (src/fir/avx.rs:25-58), finite outputs inside the gate."""
    g, _ = make_pair(2, in_hz, out_hz, lat=ra.Latency.Sample8, kernel=kernel)
    r = o.OracleFir(2, in_hz, out_hz, 16, 90, o.CONVOLVE_AVX_FMA if o.have_avx_fma() else o.CONVOLVE_SCALAR)
    n = 200_000
    x = synth.fast_noise(2 * n, seed=78)
    rng = np.random.default_rng(5)
    for k, f in enumerate(rng.choice(np.arange(100, n - 100), size=48, replace=False)):
        x[2 * f + (k & 1)] = (np.inf, -np.inf, np.nan)[k % 3]
    yg, consumed = g.resample_bulk(x, 512)
    yr, _ = r.resample_all(x, 512)
    assert consumed == x.size and yg.size == yr.size
    fin_r, fin_g = np.isfinite(yr), np.isfinite(yg)
    assert (~fin_r).any()
    assert np.array_equal(fin_g, fin_r), np.flatnonzero(fin_g != fin_r)[:10]
    assert rms(yg[fin_r], yr[fin_r]) <= RMS_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("ch,in_hz,out_hz,lat,att,frames,chunk,world", [
    (8, 96000, 44100, "Sample64", "Db120", 600_000, 512, 4),     # BASELINE config 5's stream, shortened
    (2, 44100, 48000, "Sample64", "Db90", 300_000, 512, 3),
    (1, 48000, 44100, "Sample32", "Db90", 100_003, 500, 5),
])
def test_time_sharded_stream_equals_the_single_launch(ch, in_hz, out_hz, lat, att, frames, chunk, world):
    """one long stream cut at call boundaries (sharding.fir_time_shards); every piece starts from the
    host mirror's state with its halo (ResamplerFir.seek) on a handle of its own -- as it would on its own GPU.
    Counts and states are those of the single bulk launch; the samples are the same up to the last bit or two
    (a launch premixes its coefficient rows for the position drift it sees, so two launches over different
    spans of a stream can round a coefficient differently) and within the 1e-6 RMS gate of the CPU path."""
    torch = pytest.importorskip("torch")
    from resampler_amd import sharding
    dev = torch.device("cuda:0")
    mk = lambda: ra.ResamplerFir.new_from_hz(ch, in_hz, out_hz, getattr(ra.Latency, lat), getattr(ra.Attenuation, att))
    x = synth.sweep(frames, ch, float(in_hz))
    d_x = torch.from_numpy(x).to(dev)
    whole_h = mk()
    d_whole = torch.zeros(whole_h.bulk_output_bound(x.size, chunk * ch), device=dev)
    torch.cuda.synchronize()   # (launches run on the handles' own streams: order them after the fills)
    consumed, produced = whole_h.resample_bulk_device(d_x, d_whole, chunk * ch)
    assert consumed == x.size
    shards = sharding.fir_time_shards(in_hz, out_hz, getattr(ra.Latency, lat), frames, chunk, world)
    assert sum(s.out_frames for s in shards) * ch == produced
    pieces = []
    for s in shards:
        h = mk()
        d_out = torch.zeros(max(1, s.out_frames) * ch + 64, device=dev)
        torch.cuda.synchronize()
        c, p = sharding.run_fir_time_shard(h, s, d_x, d_out, ch, chunk)
        assert (c, p) == (s.in_frames * ch, s.out_frames * ch), s.rank
        torch.cuda.synchronize()
        ref = d_whole[s.out_offset * ch:s.out_offset * ch + p]
        assert float((d_out[:p] - ref).abs().max()) <= 6e-7, s.rank          # values are below 1: a couple of ulps
        assert not d_out[p:].any()
        pieces.append(d_out[:p].cpu().numpy())
        if s.rank + 1 < world:
            assert h.state() == shards[s.rank + 1].plan.state()
        else:
            assert h.state() == whole_h.state()
    if ch * frames <= 700_000:   # the CPU path over the whole stream (kept to a few seconds)
        orc = o.OracleFir(ch, in_hz, out_hz, {"Sample32": 64, "Sample64": 128}[lat], {"Db90": 90, "Db120": 120}[att])
        y_ref, calls_ref = orc.resample_all(x, chunk * ch)
        assert int(calls_ref[:, 1].sum()) == produced and y_ref.size == produced
        assert rms(np.concatenate(pieces), y_ref) <= RMS_TOL
    # a seek with too little history is refused
    s = shards[-1]
    with pytest.raises(ra.ResampleError):
        mk().seek(s.plan, d_x[:max(0, s.history_frames - 1) * ch])


@pytest.mark.gpu
def test_three_plane_split_kernel_in_a_child_process():
    """RSMP_FIR_SPLIT_PLANES=3 (read once per process): every f32 operand cut exactly into three bf16 planes, six
    products per term -- the knob bench.py's `secondary.fir_split_bf16x3` measures; two-channel streams (the other
    channel counts keep the two-plane builds).  Same gate as the default kernel (1e-6 RMS relative to the signal,
    identical counts) on the bulk entry, interior and edge items, at full scale and at 2^-20."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import numpy as np
import resampler_amd as ra
from oracle import pyoracle as o
from resampler_amd import synth
kind = o.CONVOLVE_AVX_FMA if o.have_avx_fma() else o.CONVOLVE_SCALAR
for ch, a, b, n, level in ((2, 44100, 48000, 300000, 1.0), (2, 48000, 44100, 200000, 2.0 ** -20), (8, 44100, 48000, 120000, 1.0),
                           (1, 48000, 44100, 150000, 1.0), (3, 44100, 48000, 100000, 30.0)):
    g = ra.ResamplerFir.new_from_hz(ch, a, b, ra.Latency.Sample64, ra.Attenuation.Db90)
    g.set_kernel(ra.FirKernel.Periodic)
    r = o.OracleFir(ch, a, b, 128, 90, kind)
    x = (synth.sweep(n, ch, float(a)) * np.float32(level)).astype(np.float32)
    chunk = 512 - 512 % ch
    yg, consumed, calls_g = g.resample_bulk(x, chunk, want_calls=True)
    yr, calls_r = r.resample_all(x, chunk)
    assert g.kernel_variant() == (4 if ch == 2 else 5), (ch, a, b, g.kernel_variant())
    assert yg.size == yr.size and np.array_equal(calls_g, calls_r), (ch, a, b)
    e = float(np.sqrt(np.mean((yg.astype(np.float64) - yr) ** 2))) / float(np.sqrt(np.mean(yr.astype(np.float64) ** 2)))
    assert e <= 1e-6, (ch, a, b, e)
print("three planes ok")
"""
    env = dict(os.environ, RSMP_DEBUG="1", RSMP_FIR_SPLIT_PLANES="3", PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "three planes ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("ch,in_hz,out_hz,latency,frames", [
    (2, 44100, 47999, "Sample64", 90000), (1, 44100, 47999, "Sample64", 70000), (3, 48000, 44101, "Sample64", 50000),
    (2, 96000, 44101, "Sample64", 120000), (8, 44100, 47999, "Sample64", 30000), (2, 44101, 47999, "Sample16", 60000),
    (2, 47999, 44100, "Sample32", 60000), (5, 44100, 47999, "Sample8", 40000)])
def test_long_launches_of_ratios_without_a_short_period(ch, in_hz, out_hz, latency, frames):
    """ResamplerFir::new_from_hz with arbitrary rates (src/resampler_fir.rs:295-301, tested at :841-862): a ratio like 44100 / 47999 has
    no short period, every output has a phase row of its own.  Long launches of such streams take fir_generic_bulk.hip -- tiles of up to
    4096 outputs sorted by phase row, the tile's window in LDS --: counts identical to the reference's call loop,
    samples within 1e-6 RMS of the AVX+FMA oracle relative to the signal; 1 / 2 / odd / many channels, 16 .. 128 taps, both directions.
    The launch's end state goes on into a second launch (its history buffer is what the first one left)."""
    lat = getattr(ra.Latency, latency)
    kind = o.CONVOLVE_AVX_FMA if o.have_avx_fma() else o.CONVOLVE_SCALAR
    gpu = ra.ResamplerFir.new_from_hz(ch, in_hz, out_hz, lat, ra.Attenuation.Db90)
    ref = o.OracleFir(ch, in_hz, out_hz, 2 * int(latency[len("Sample"):]), 90, kind)
    x = synth.sweep(frames, ch, float(in_hz))
    chunk = 512 - 512 % ch
    half = (frames // 2) * ch
    for part in (x[:half], x[half:]):
        yg, consumed, calls_g = gpu.resample_bulk(part, chunk, want_calls=True)
        yr, calls_r = ref.resample_all(part, chunk)
        assert gpu.kernel_variant() == 0, gpu.kernel_variant()   # (no periodic kernel takes these ratios)
        assert yg.size == yr.size and np.array_equal(calls_g, calls_r)
        e = float(np.sqrt(np.mean((yg.astype(np.float64) - yr) ** 2))) / float(np.sqrt(np.mean(yr.astype(np.float64) ** 2)))
        assert e <= 1e-6, (ch, in_hz, out_hz, e)


@pytest.mark.gpu
def test_clean_audio_needs_no_repair_pass_in_a_child_process():
    """The repair pass (fir_nonfinite.h) exists for inf / NaN / out-of-range samples and for items whose predicted scale was
    off; clean audio must come out of the periodic kernels right BY ITSELF.  Round 5 shipped odd channel counts that did
    not: the phantom channel of an odd count's last pair was the next frame's first channel, scaled from a history of silence
    behind an edge item -- infinities in its planes, NaN (0 x inf) in the last class tiles of the image one slot before, one
    item per workgroup range redone by the repair pass (three channels: 1.0 ms of repairs on a 0.68 ms launch), and no test
    noticed because the repaired output is right.  RSMP_FIR_NO_REPAIR=1 (debug knob, read once per process) drops the repair
    launch: noise at a level per stream, several streams per launch (stream edges inside workgroup ranges), every channel
    count's kind of pair -- against the oracle, every stream."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import numpy as np
import torch
import resampler_amd as ra
from oracle import pyoracle as o
from resampler_amd import synth
kind = o.CONVOLVE_AVX_FMA if o.have_avx_fma() else o.CONVOLVE_SCALAR
dev = torch.device("cuda:0")
for ch, a, b, streams in ((3, 44100, 48000, 5), (5, 48000, 44100, 3), (1, 48000, 44100, 6), (2, 44100, 48000, 6), (4, 44100, 48000, 4), (7, 44100, 48000, 2)):
    frames = 60000
    hs = [ra.ResamplerFir.new_from_hz(ch, a, b, ra.Latency.Sample64, ra.Attenuation.Db90) for _ in range(streams)]
    x = synth.fast_noise(frames * ch, seed=7 + ch)
    xs = [(x * np.float32(0.25 + 0.75 * i / streams)).astype(np.float32) for i in range(streams)]
    d_in = [torch.from_numpy(v).to(dev) for v in xs]
    d_out = [torch.zeros(hs[0].bulk_output_bound(frames * ch, 512 * ch), device=dev) for _ in hs]
    batch = ra.FirBatch(hs)
    batch.bind(d_in, d_out)
    cons, prod = batch.resample_bulk_device(512 * ch, ra.torch_stream())
    torch.cuda.synchronize()
    assert hs[0].kernel_variant() in (4, 5), (ch, hs[0].kernel_variant())
    for i in range(streams):
        r = o.OracleFir(ch, a, b, 128, 90, kind)
        yr, _ = r.resample_all(xs[i], 512 * ch)
        yg = d_out[i].cpu().numpy()[:int(prod[i])]
        assert yg.size == yr.size, (ch, i, yg.size, yr.size)
        g2, r2 = yg.reshape(-1, ch).astype(np.float64), yr.reshape(-1, ch).astype(np.float64)
        for c in range(ch):
            e = float(np.sqrt(np.mean(np.nan_to_num(g2[:, c] - r2[:, c], nan=1.0) ** 2))) / float(np.sqrt(np.mean(r2[:, c] ** 2)))
            assert e <= 1e-6, (ch, a, b, i, c, e)
print("no repair needed")
"""
    env = dict(os.environ, RSMP_DEBUG="1", RSMP_FIR_NO_REPAIR="1", PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "no repair needed" in out.stdout, out.stdout + out.stderr


PAIR_FLOOR = 2.0 ** -36   # what a sample keeps at least, relative to the largest sample of its channel pair's work item


def _per_channel_errors(yg, yr, ch):
    yg, yr = yg.reshape(-1, ch).astype(np.float64), yr.reshape(-1, ch).astype(np.float64)
    return [(float(np.sqrt(np.mean((yg[:, c] - yr[:, c]) ** 2))), float(np.sqrt(np.mean(yr[:, c] ** 2)))) for c in range(ch)]


@pytest.mark.parametrize("ch,in_hz,out_hz", [(2, 44100, 48000), (4, 44100, 48000), (8, 48000, 44100), (8, 96000, 44100),
                                             (2, 96000, 44100), (2, 44100, 96000)])
@pytest.mark.parametrize("quiet_exp", [12, 20, 30])
def test_channels_of_a_pair_at_very_different_levels(ch, in_hz, out_hz, quiet_exp):
    """The reference computes every channel on its own (src/resampler_fir.rs:567-586); the split kernels cut a channel
    PAIR's samples into fp16 planes per work item (16 periods of two channels) -- with a block-floating-point scale PER
    CHANNEL since round 5 (rounds 3-4: one per pair, which left a channel 2^-20 below its partner with 8e-7 .. 9e-6 of
    its own level and this test with a documented floor).  The odd channels here carry the sweep at 2^-12, 2^-20, 2^-30
    of the even ones: EVERY channel keeps 1e-6 RMS relative to its own level."""
    g, r = make_pair(ch, in_hz, out_hz, kernel=ra.FirKernel.Periodic)
    n = 70000
    x = synth.sweep(n, ch, float(in_hz)).reshape(n, ch).copy()
    x[:, 1::2] *= np.float32(2.0 ** -quiet_exp)
    x = x.reshape(-1)
    yg, _ = g.resample_bulk(x, 512 - 512 % ch)
    yr, _ = r.resample_all(x, 512 - 512 % ch)
    assert yg.size == yr.size
    for c, (err, level) in enumerate(_per_channel_errors(yg, yr, ch)):
        assert err <= RMS_TOL * level, (c, err, level)


def test_a_silent_channel_beside_a_loud_one():
    """Digital silence in one channel of a pair (a mono recording in a stereo file): its outputs are exact zeros, its
    partner keeps 1e-6 of its level, and nothing is sent to the repair pass for it (the silent channel has no scale to
    be off)."""
    g, r = make_pair(2, 44100, 48000, kernel=ra.FirKernel.Periodic)
    n = 200000
    x = synth.sweep(n, 2, 44100.0).reshape(n, 2).copy()
    x[:, 1] = 0.0
    x = x.reshape(-1)
    yg, _ = g.resample_bulk(x, 512)
    yr, _ = r.resample_all(x, 512)
    assert yg.size == yr.size
    (e0, l0), (e1, l1) = _per_channel_errors(yg, yr, 2)
    assert e0 <= RMS_TOL * l0, (e0, l0)
    assert l1 == 0.0 and e1 == 0.0, (e1, l1)


def test_a_burst_followed_by_near_silence_inside_one_work_item():
    """Full-scale samples for 40 frames, 2^-20 noise around them: the outputs behind the burst that still belong to its
    16-period work item (53 ms at 44.1 kHz) are scaled for the burst.  They keep 1e-6 of THEIR level here; the documented
    floor is 2^-36 of the burst."""
    g, r = make_pair(2, 44100, 48000, kernel=ra.FirKernel.Periodic)
    n = 70000
    x = (synth.fast_noise(n * 2, seed=3) * np.float32(2.0 ** -20)).reshape(n, 2)
    x[1000:1040] = synth.fast_noise(80, seed=4).reshape(40, 2)
    x = x.reshape(-1).astype(np.float32)
    yg, _ = g.resample_bulk(x, 512)
    yr, _ = r.resample_all(x, 512)
    assert rel_rms(yg, yr) <= RMS_TOL
    seg = slice(2 * 1400, 2 * 3400)   # behind the burst, inside its work item
    err = rms(yg[seg], yr[seg])
    level = float(np.sqrt(np.mean(yr[seg].astype(np.float64) ** 2)))
    assert err <= max(RMS_TOL * level, PAIR_FLOOR), (err, level)


def test_lockstep_channels_at_very_different_levels():
    """The same through the lock-step batch (a scale per stream, step AND channel there) and through runs of several steps
    (the bulk kernels): every channel 1e-6 of its own level."""
    torch = pytest.importorskip("torch")
    from resampler_amd import sharding
    dev = torch.device("cuda:0")
    specs = sharding.mixed_rate_batch(12, 2, 512)
    steps = 12
    for mode in ("step", "run"):
        hs = [ra.ResamplerFir.new_from_hz(2, s.in_hz, s.out_hz, ra.Latency.Sample64, ra.Attenuation.Db90) for s in specs]
        refs = [o.OracleFir(2, s.in_hz, s.out_hz, 128, 90, ORACLE_KIND) for s in specs]
        xs = []
        for i, s in enumerate(specs):
            x = synth.sweep(steps * 512, 2, float(s.in_hz)).reshape(-1, 2).copy()
            x[:, 1] *= np.float32(2.0 ** -(12, 20, 30)[(i // 6) % 3 if i >= 6 else 1])
            xs.append(x.reshape(-1))
        caps = [h.buffer_size_output() for h in hs]
        d_in = [torch.from_numpy(x).to(dev) for x in xs]
        d_out = [torch.zeros(steps * c, device=dev) for c in caps]
        ls = ra.FirLockstep(hs, 512)
        ls.bind_caps(d_in, d_out, caps)
        if mode == "run":
            ls.run(steps, 512, 0, append=True)
        else:
            for k in range(steps):
                ls.step(512, k * 512, append=True)
        ls.sync()
        for i, r in enumerate(refs):
            yr, _ = r.resample_all(xs[i], 1024)
            yg = d_out[i][:yr.size].cpu().numpy()
            for c, (err, level) in enumerate(_per_channel_errors(yg, yr, 2)):
                assert err <= RMS_TOL * level, (mode, i, c, err, level)
        ls.close()


def test_a_bulk_call_longer_than_one_launch():
    """50 M frames in one bulk call (19 minutes at 44.1 kHz): more outputs than one launch takes (its coefficient rows
    are mixed for one drift; the library cuts the call at call boundaries, fir_api.cpp run_single) -- the calls, the samples
    (over the whole stream and over its first and last twentieth, where a table mixed for the wrong end would show) and
    the final state are the reference's."""
    if not o.have_avx_fma():
        pytest.skip("needs the oracle's AVX + FMA path (time)")
    n = 50_000_000
    x = synth.fast_noise(2 * n, seed=41)
    g = ra.ResamplerFir.new_from_hz(2, 44100, 48000, ra.Latency.Sample64, ra.Attenuation.Db90)
    r = o.OracleFir(2, 44100, 48000, 128, 90, o.CONVOLVE_AVX_FMA)
    yg, consumed, calls_g = g.resample_bulk(x, 1024, want_calls=True)
    yr, calls_r = r.resample_all(x, 1024)
    assert consumed == x.size and np.array_equal(calls_g, calls_r)
    assert yg.size == yr.size
    part = yr.size // 20
    for a, b in ((0, part), (yr.size - part, yr.size), (0, yr.size)):
        d = yg[a:b].astype(np.float64) - yr[a:b]
        assert np.sqrt(np.dot(d, d) / (b - a)) <= RMS_TOL, (a, b)
    assert g.state() == r.state()


def test_a_batch_with_one_stream_longer_than_one_launch_and_mixed_rate_pairs():
    """A bulk batch of three streams: 44.1 -> 48 kHz with 50 M frames (cut into two launches, the others take part in the
    first only), 48 -> 44.1 kHz and 96 -> 44.1 kHz with 2^20 (three rate pairs: the split kernel's jobs share launches).
    Counts, samples and states per stream are the reference's."""
    import torch
    if not o.have_avx_fma():
        pytest.skip("needs the oracle's AVX + FMA path (time)")
    dev = torch.device("cuda:0")
    specs = [(44100, 48000, 50_000_000), (48000, 44100, 1 << 20), (96000, 44100, 1 << 20)]
    hs = [ra.ResamplerFir.new_from_hz(2, i, o_, ra.Latency.Sample64, ra.Attenuation.Db90) for i, o_, _ in specs]
    refs = [o.OracleFir(2, i, o_, 128, 90, o.CONVOLVE_AVX_FMA) for i, o_, _ in specs]
    xs = [synth.fast_noise(2 * n, seed=50 + k) for k, (_, _, n) in enumerate(specs)]
    d_in = [torch.from_numpy(x).to(dev) for x in xs]
    d_out = [torch.empty(h.bulk_output_bound(x.size, 1024), device=dev) for h, x in zip(hs, xs)]
    b = ra.FirBatch(hs)
    b.bind(d_in, d_out)
    consumed, produced = b.resample_bulk_device(1024, ra.torch_stream())
    torch.cuda.synchronize()
    for k, (h, r, x) in enumerate(zip(hs, refs, xs)):
        want, _ = r.resample_all(x, 1024)
        assert consumed[k] == x.size and produced[k] == want.size, k
        got = d_out[k][:want.size].cpu().numpy()
        part = max(1, want.size // 20)
        for a, e in ((0, part), (want.size - part, want.size), (0, want.size)):
            d = got[a:e].astype(np.float64) - want[a:e]
            assert np.sqrt(np.dot(d, d) / (e - a)) <= RMS_TOL, (k, a, e)
        assert h.state() == r.state(), k


@pytest.mark.gpu
def test_a_batch_in_many_states_goes_through_the_device_planner_and_back():
    """FirBatch.resample_bulk_device routes a batch whose streams are in many different states through the device planner
    (rsmp_fir_lockstep_run_bulk over the same handles, states written back before the call returns) and everything else
    through the host planner -- whichever it takes, and however the two and a stream's own entries alternate on the same
    handles: the reference driver loop's (consumed, produced), its samples, its end states."""
    import torch
    dev = torch.device("cuda:0")
    n, chunk = 20, 256
    pairs = [(44100, 48000), (48000, 44100), (96000, 44100), (44100, 96000)]
    hs = [ra.ResamplerFir.new_from_hz(2, *pairs[i % 4], ra.Latency.Sample64, ra.Attenuation.Db90) for i in range(n)]
    kind = o.CONVOLVE_AVX_FMA if o.have_avx_fma() else o.CONVOLVE_SCALAR
    refs = [o.OracleFir(2, *pairs[i % 4], 128, 90, kind) for i in range(n)]
    rng = np.random.default_rng(5)

    def feed_own_entry(i, frames):   # a stream through its own resample(): the handle moves on under any batch
        x = (rng.random(2 * frames, dtype=np.float32) * 2 - 1).astype(np.float32)
        og, orr = np.zeros(hs[i].buffer_size_output(), np.float32), np.zeros(refs[i].buffer_size_output(), np.float32)
        off = 0
        while off < x.size:
            cg, pg = hs[i].resample(x[off:off + 2 * 150], og)
            rc, cr, pr = refs[i].resample(x[off:off + 2 * 150], orr)
            assert rc == 0 and (cg, pg) == (cr, pr)
            assert rms(og[:pg], orr[:pr]) <= RMS_TOL
            off += cg

    for i in range(n):
        feed_own_entry(i, 200 + 31 * i)   # twenty states
    batch = ra.FirBatch(hs)
    frames = 30000
    routed = []
    for launch in range(6):
        if launch == 3:
            feed_own_entry(7, 333)        # behind the batch's back: the library's lock-step batch is out of sync and is made anew
        if launch == 4:
            batch.device_planner = False  # ... and the host planner on the handles the device planner left
        if launch == 5:
            batch.device_planner = None
        xs = [(rng.random(2 * frames, dtype=np.float32) * 2 - 1).astype(np.float32) for _ in range(n)]
        d_in = [torch.from_numpy(x).to(dev) for x in xs]
        d_out = [torch.zeros(h.bulk_output_bound(2 * frames, chunk), device=dev) for h in hs]
        batch.bind(d_in, d_out)
        cons, prod = batch.resample_bulk_device(chunk)
        torch.cuda.synchronize()
        routed.append(batch.planned_on_device)
        for i, r in enumerate(refs):
            y, calls = r.resample_all(xs[i], chunk)
            assert int(cons[i]) == xs[i].size and int(prod[i]) == y.size, (launch, i, int(cons[i]), int(prod[i]), y.size)
            assert rms(d_out[i][:y.size].cpu().numpy(), y) <= RMS_TOL, (launch, i)
            assert hs[i].state() == r.state(), (launch, i)
    assert routed == [True, True, True, True, False, True], routed
    # all streams in ONE state: the host planner's shared plan, no device planning
    for h, r in zip(hs, refs):
        h.reset()
        r.reset()
    same = ra.FirBatch(hs[0::4])
    xs = [(rng.random(2 * frames, dtype=np.float32) * 2 - 1).astype(np.float32)] * len(hs[0::4])
    d_in = [torch.from_numpy(x).to(dev) for x in xs]
    d_out = [torch.zeros(h.bulk_output_bound(2 * frames, chunk), device=dev) for h in hs[0::4]]
    same.bind(d_in, d_out)
    cons, prod = same.resample_bulk_device(chunk)
    torch.cuda.synchronize()
    assert not same.planned_on_device
    y, _ = refs[0].resample_all(xs[0], chunk)
    assert int(prod[0]) == y.size and rms(d_out[0][:y.size].cpu().numpy(), y) <= RMS_TOL



@pytest.mark.gpu
def test_routed_batches_survive_their_handles():
    """The library keeps the lock-step batch of a routed launch per list of handles: freeing the handles drops it (no write-back
    into freed memory, no stale entry for handles that come back at the same addresses), and more handle lists than the cache
    holds push the oldest out."""
    import gc
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(9)
    for round_ in range(12):   # (the cache holds eight handle lists)
        n = 16
        hs = [ra.ResamplerFir.new_from_hz(2, 44100, 48000, ra.Latency.Sample32, ra.Attenuation.Db90) for _ in range(n)]
        refs = [o.OracleFir(2, 44100, 48000, hs[0].taps, 90) for _ in range(n)]
        for i in range(n):   # sixteen states
            x = (rng.random(2 * (100 + 17 * i), dtype=np.float32) * 2 - 1).astype(np.float32)
            og, orr = np.zeros(hs[i].buffer_size_output(), np.float32), np.zeros(refs[i].buffer_size_output(), np.float32)
            cg, pg = hs[i].resample(x, og)
            rc, cr, pr = refs[i].resample(x, orr)
            assert rc == 0 and (cg, pg) == (cr, pr)
        frames, chunk = 6000, 256
        xs = [(rng.random(2 * frames, dtype=np.float32) * 2 - 1).astype(np.float32) for _ in range(n)]
        d_in = [torch.from_numpy(x).to(dev) for x in xs]
        d_out = [torch.zeros(h.bulk_output_bound(2 * frames, chunk), device=dev) for h in hs]
        batch = ra.FirBatch(hs)
        batch.bind(d_in, d_out)
        for launch in range(2):
            cons, prod = batch.resample_bulk_device(chunk)
            assert batch.planned_on_device
            for i in (0, 7, 15):
                y, _ = refs[i].resample_all(xs[i], chunk) if launch == 0 or True else (None, None)
                assert int(prod[i]) == y.size and rms(d_out[i][:y.size].cpu().numpy(), y) <= RMS_TOL, (round_, launch, i)
            for i in range(n):
                if i not in (0, 7, 15):
                    refs[i].resample_all(xs[i], chunk)
            for h, r in zip(hs, refs):
                assert h.state() == r.state()
        if round_ % 2:
            del batch, hs   # (freed: the library's batch over them goes too)
            gc.collect()
