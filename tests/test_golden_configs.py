"""Golden fixtures of the five BASELINE configs (tests/golden/config_fixtures.json, written by
tests/golden/make_fixtures.py from the CPU oracle in the build container).  The reference pins no
end-to-end sample or count sequence, so these freeze the oracle: a box whose libm moves the
design tables, or an oracle edit that changes a rounding, fails here -- and the HIP path is checked against
the frozen values, not only against an oracle running beside it."""
import base64
import hashlib
import json
import os

import numpy as np
import pytest

import resampler_amd as ra
from oracle import pyoracle as o
from resampler_amd import sharding, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RMS_TOL = 1e-6   # north_star tolerance


@pytest.fixture(scope="module")
def fx():
    with open(os.path.join(ROOT, "tests", "golden", "config_fixtures.json")) as f:
        return json.load(f)


def unpack(s):
    return np.frombuffer(base64.b64decode(s), "<f4")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, np.float32).tobytes()).hexdigest()


def rms(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


def expand(rle):
    out = []
    for c, p, k in rle:
        out += [[c, p]] * k
    return np.array(out, np.int64)


def fir_input(name):
    if name == "c1":
        return synth.sweep(1 << 20, 1, 48000.0)
    if name.startswith("c2"):
        return synth.sweep(1 << 20, 2, 44100.0)
    return synth.hash_noise(64 * 512 * 8, seed=5)


def check_oracle(fx):
    names = ("c1", "c2", "c5") + (("c2_avx_fma", "c5_avx_fma") if o.have_avx_fma() else ())
    for name in names:
        c = fx[name]
        kind = o.CONVOLVE_AVX_FMA if name.endswith("avx_fma") else o.CONVOLVE_SCALAR
        r = o.OracleFir(c["channels"], c["in_hz"], c["out_hz"], c["taps"], c["attenuation_db"], kind)
        assert sha(r.coeffs()) == c["table_sha256"], name       # the polyphase table, bit for bit
        y, calls = r.resample_all(fir_input(name), c["chunk_values"])
        assert y.size == c["out_values"] and np.array_equal(calls, expand(c["calls_rle"])), name
        assert sha(y) == c["sha256"], name
        assert list(r.state()) == c["final_state"], name
    c3 = fx["c3"]
    f = o.OracleFft(2, 44100, 48000)
    assert sha(f.filter_spectrum().view(np.float32)) == c3["filter_spectrum_sha256"]
    n_in, n_out = f.chunk_size_input(), f.chunk_size_output()
    assert (n_in, n_out) == (c3["chunk_size_input"], c3["chunk_size_output"])
    x = synth.sweep(c3["blocks"] * n_in // 2, 2, 44100.0)
    y = np.zeros(c3["blocks"] * n_out, np.float32)
    blk = np.zeros(n_out, np.float32)
    for b in range(c3["blocks"]):
        assert f.resample(x[b * n_in:(b + 1) * n_in], blk) == 0
        y[b * n_out:(b + 1) * n_out] = blk
    assert sha(y) == c3["sha256"]
    c4 = fx["c4"]
    specs = sharding.mixed_rate_batch(c4["streams"], 2, c4["frames_per_step"])
    all_hash = hashlib.sha256()
    detail = {d["index"]: d for d in c4["detail"]}
    for i, s in enumerate(specs):
        r = o.OracleFir(2, s.in_hz, s.out_hz, 128, 90)
        x = synth.hash_noise(c4["steps"] * 1024, seed=i)
        out = np.zeros(r.buffer_size_output(), np.float32)
        ys, counts = [], []
        for k in range(c4["steps"]):
            rc, c, p = r.resample(x[k * 1024:(k + 1) * 1024], out)
            assert rc == 0
            counts.append([c, p])
            ys.append(out[:p].copy())
        digest = sha(np.concatenate(ys))
        all_hash.update(bytes.fromhex(digest))
        if i in detail:
            assert counts == detail[i]["counts"] and digest == detail[i]["sha256"], i
    assert all_hash.hexdigest() == c4["sha256_of_stream_sha256s"]
    if o.have_avx_fma():   # config 4 at its full length (256 steps), two streams of every rate pair, AVX + FMA path
        c4l = fx["c4_256"]
        for d in c4l["detail"]:
            r = o.OracleFir(2, d["in_hz"], d["out_hz"], 128, 90, o.CONVOLVE_AVX_FMA)
            x = synth.hash_noise(c4l["steps"] * 1024, seed=d["index"])
            out = np.zeros(r.buffer_size_output(), np.float32)
            ys, counts = [], []
            for k in range(c4l["steps"]):
                rc, c, p = r.resample(x[k * 1024:(k + 1) * 1024], out)
                assert rc == 0
                counts.append([c, p])
                ys.append(out[:p].copy())
            assert np.array_equal(np.array(counts), expand(d["counts_rle"])), d["index"]
            assert sha(np.concatenate(ys)) == d["sha256"] and list(r.state()) == d["final_state"], d["index"]


def test_oracle_reproduces_the_golden_configs(fx):
    check_oracle(fx)


def test_product_tables_match_the_golden_hashes(fx):
    # host-only design path of the product (no GPU): the same 128-tap tables, bit for bit
    for name, att in (("c1", ra.Attenuation.Db90), ("c2", ra.Attenuation.Db90), ("c5", ra.Attenuation.Db120)):
        c = fx[name]
        t = ra.design_fir_coeffs(c["in_hz"], c["out_hz"], ra.Latency.Sample64, att)
        assert sha(t) == c["table_sha256"], name


@pytest.mark.gpu
def test_oracle_reproduces_the_golden_configs_on_the_gpu_box(fx):
    check_oracle(fx)


@pytest.mark.gpu
def test_hip_path_against_the_golden_configs(fx):
    att = {90: ra.Attenuation.Db90, 120: ra.Attenuation.Db120}
    # C1 / C2 / C5 through the bulk driver (the reference loop with the config's call size)
    for name in ("c1", "c2", "c5", "c2_avx_fma", "c5_avx_fma"):   # (*_avx_fma: the CPU SIMD path north_star names)
        c = fx[name]
        g = ra.ResamplerFir.new_from_hz(c["channels"], c["in_hz"], c["out_hz"], ra.Latency.Sample64, att[c["attenuation_db"]])
        assert g.buffer_size_output() == c["buffer_size_output"]
        y, consumed, calls = g.resample_bulk(fir_input(name), c["chunk_values"], want_calls=True)
        assert consumed == c["in_values"] and y.size == c["out_values"], name
        assert np.array_equal(calls, expand(c["calls_rle"])), name
        assert list(g.state()) == c["final_state"], name
        head, tail = unpack(c["head"]), unpack(c["tail"])
        assert rms(y[:head.size], head) <= RMS_TOL and rms(y[-tail.size:], tail) <= RMS_TOL, name
    # C1 also call by call (the drop-in form): counts of every call
    c = fx["c1"]
    g = ra.ResamplerFir.new_from_hz(1, 48000, 44100, ra.Latency.Sample64, ra.Attenuation.Db90)
    x = fir_input("c1")
    out = np.zeros(g.buffer_size_output(), np.float32)
    want = expand(c["calls_rle"])
    got_head = []
    for k in range(64):
        cg, pg = g.resample(x[k * 512:(k + 1) * 512], out)
        assert [cg, pg] == want[k].tolist(), k
        got_head.append(out[:pg].copy())
    got_head = np.concatenate(got_head)
    head = unpack(c["head"])
    m = min(head.size, got_head.size)
    assert rms(got_head[:m], head[:m]) <= RMS_TOL
    # C3
    c3 = fx["c3"]
    f = ra.ResamplerFft.new(2, ra.SampleRate.Hz44100, ra.SampleRate.Hz48000)
    x = synth.sweep(c3["blocks"] * c3["chunk_size_input"] // 2, 2, 44100.0)
    y = f.resample_bulk(x, c3["blocks"])
    head, tail = unpack(c3["head"]), unpack(c3["tail"])
    assert rms(y[:head.size], head) <= RMS_TOL and rms(y[-tail.size:], tail) <= RMS_TOL
    # C4: the lock-step batch at the config's full shape
    import torch
    dev = torch.device("cuda:0")
    c4 = fx["c4"]
    specs = sharding.mixed_rate_batch(c4["streams"], 2, c4["frames_per_step"])
    hs = [ra.ResamplerFir.new_from_hz(2, s.in_hz, s.out_hz, ra.Latency.Sample64, ra.Attenuation.Db90) for s in specs]
    d_in = [torch.from_numpy(synth.hash_noise(c4["steps"] * 1024, seed=i)).to(dev) for i in range(len(specs))]
    caps = [h.buffer_size_output() for h in hs]
    d_out = [torch.zeros(c4["steps"] * cap, device=dev) for cap in caps]
    ls = ra.FirLockstep(hs, c4["frames_per_step"])
    ls.bind_caps(d_in, d_out, caps)
    detail = {d["index"]: d for d in c4["detail"]}
    counts = {i: [] for i in detail}
    for k in range(c4["steps"]):
        ls.step(c4["frames_per_step"], k * c4["frames_per_step"], append=True)
        cons, prod = ls.counts()
        for i in detail:
            counts[i].append([int(cons[i]), int(prod[i])])
    for i, d in detail.items():
        assert counts[i] == d["counts"], i
        total = sum(p for _, p in d["counts"])
        y = d_out[i][:total].cpu().numpy()
        head, tail = unpack(d["head"]), unpack(d["tail"])
        assert rms(y[:head.size], head) <= RMS_TOL and rms(y[-tail.size:], tail) <= RMS_TOL, i
    ls.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 16, 256])
def test_config4_at_its_full_length(fx, k):
    """BASELINE config 4 as stated: 1024 mixed-rate streams x 256 lock-step steps of 512 frames, k steps per launch (k = 1:
    rsmp_fir_lockstep_step; k = 16, 256: rsmp_fir_lockstep_run -- planned on the device, one bulk launch per rate pair;
    k = 256 is the whole configuration in one go).  Every stream's
    (consumed, produced) of every step against the host mirror of the reference state machine (exact, no samples);
    samples, counts and the final state of two streams per rate pair against the frozen AVX+FMA oracle run
    (fixture c4_256), and of 18 more streams spread over the batch against the oracle run here."""
    import torch
    dev = torch.device("cuda:0")
    c4l = fx["c4_256"]
    n, steps, frames = 1024, c4l["steps"], c4l["frames_per_step"]
    specs = sharding.mixed_rate_batch(n, 2, frames)
    hs = [ra.ResamplerFir.new_from_hz(2, s.in_hz, s.out_hz, ra.Latency.Sample64, ra.Attenuation.Db90) for s in specs]
    checked = sorted(set(d["index"] for d in c4l["detail"]) | set(range(100, 1024, 53)))
    caps = [h.buffer_size_output() for h in hs]
    # every stream reads its own input and appends its outputs (at most ceil(512 * out / in) + 1 frames per step)
    d_in = [torch.from_numpy(synth.hash_noise(steps * frames * 2, seed=i)).to(dev) for i in range(n)]
    room = [steps * 2 * (frames * s.out_hz // s.in_hz + 2) + caps[i] for i, s in enumerate(specs)]
    d_out = [torch.zeros(room[i], device=dev) for i in range(n)]
    plans = [ra.FirPlan(s.in_hz, s.out_hz, ra.Latency.Sample64) for s in specs[:6]]   # one per rate pair
    want_counts = []
    for i, pl in enumerate(plans):
        per = []
        for _ in range(steps):
            acc, prod = pl.call(frames, caps[i] // 2)
            per.append((acc * 2, prod * 2))
        want_counts.append(per)
    ls = ra.FirLockstep(hs, frames)
    ls.bind_caps(d_in, d_out, caps)
    got = {i: [] for i in checked}
    want = np.array([[want_counts[i % 6][s] for i in range(n)] for s in range(steps)])   # [step][stream][2]
    for s0 in range(0, steps, k):
        if k == 1:
            ls.step(frames, s0 * frames, append=True)
            cons, prod = ls.counts()
            cons, prod = cons[None, :], prod[None, :]
        else:
            ls.run(k, frames, s0 * frames, append=True)
            cons, prod = ls.run_counts()
            assert ls.run_slow_calls() == 0   # (these six rate pairs never leave the planner's fast path)
        assert np.array_equal(cons, want[s0:s0 + k, :, 0]) and np.array_equal(prod, want[s0:s0 + k, :, 1]), s0
        for i in checked:
            got[i].extend([[int(cons[s][i]), int(prod[s][i])] for s in range(cons.shape[0])])
    assert not ls.status().any()
    detail = {d["index"]: d for d in c4l["detail"]}
    for i in checked:
        total = sum(p for _, p in got[i])
        y = d_out[i][:total].cpu().numpy()
        if i in detail:
            d = detail[i]
            assert np.array_equal(np.array(got[i]), expand(d["counts_rle"])), i
            head, tail = unpack(d["head"]), unpack(d["tail"])
            assert rms(y[:head.size], head) <= RMS_TOL and rms(y[-tail.size:], tail) <= RMS_TOL, i
        r = o.OracleFir(2, specs[i].in_hz, specs[i].out_hz, 128, 90, o.CONVOLVE_AVX_FMA if o.have_avx_fma() else o.CONVOLVE_SCALAR)
        x = synth.hash_noise(steps * frames * 2, seed=i)
        out = np.zeros(caps[i], np.float32)
        ys = []
        for s in range(steps):
            rc, c, p = r.resample(x[s * 1024:(s + 1) * 1024], out)
            assert rc == 0 and [c, p] == got[i][s], (i, s)
            ys.append(out[:p].copy())
        assert rms(y, np.concatenate(ys)) <= RMS_TOL, i
    # the states written back equal the oracle's, bit for bit
    ls.sync()
    for i in checked:
        r = o.OracleFir(2, specs[i].in_hz, specs[i].out_hz, 128, 90)
        out = np.zeros(caps[i], np.float32)
        x = synth.hash_noise(steps * frames * 2, seed=i)
        for s in range(steps):
            r.resample(x[s * 1024:(s + 1) * 1024], out)
        assert hs[i].state() == r.state(), i
    ls.close()


@pytest.mark.gpu
def test_config5_at_its_full_length(fx):
    """BASELINE config 5 as stated: ONE 8-channel 96 -> 44.1 kHz stream of ten minutes (57.6 M frames, 128 taps, Db120,
    512-frame calls), sample by sample: the oracle's convolve_interp_avx_fma run must reproduce the frozen hashes (fixture
    c5_full: output, per-call counts, final state), the HIP path's bulk launch must return the same counts for all
    112 500 calls, the same final state, and every output value within 1e-6 RMS over the whole stream and over each
    tenth of it (the channel-pair kernel rebuilds its class table as the f64 position drifts: the end of the stream is
    where that would show)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_fixtures import c5_full_input
    if not o.have_avx_fma():
        pytest.skip("the frozen run is the AVX + FMA path")
    c5 = fx["c5_full"]
    x = c5_full_input()
    assert x.size == c5["in_values"]
    ref = o.OracleFir(8, 96000, 44100, 128, 120, o.CONVOLVE_AVX_FMA)
    want, calls = ref.resample_all(x, c5["chunk_values"])
    assert want.size == c5["out_values"] and calls.shape[0] == c5["n_calls"]
    assert hashlib.sha256(np.ascontiguousarray(calls, "<i8").tobytes()).hexdigest() == c5["calls_sha256"]
    assert sha(want) == c5["sha256"]
    assert list(ref.state()) == c5["final_state"]
    gpu = ra.ResamplerFir.new_from_hz(8, 96000, 44100, ra.Latency.Sample64, ra.Attenuation.Db120)
    got, consumed, g_calls = gpu.resample_bulk(x, c5["chunk_values"], want_calls=True)
    assert consumed == x.size and got.size == want.size
    assert np.array_equal(g_calls, calls)
    assert list(gpu.state()) == c5["final_state"]
    tenth = want.size // 10
    total = 0.0
    for part in range(10):
        a, b = part * tenth, (want.size if part == 9 else (part + 1) * tenth)
        d = got[a:b].astype(np.float64) - want[a:b]
        sq = float(np.dot(d, d))
        assert np.sqrt(sq / (b - a)) <= RMS_TOL, (part, np.sqrt(sq / (b - a)))
        total += sq
    assert np.sqrt(total / want.size) <= RMS_TOL
