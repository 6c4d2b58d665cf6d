"""tools/soak_lockstep.py [--hours H | runs] [k] -- a long lock-step session on one small mixed batch.

Default (no --hours): thousands of runs of k calls (the next one planned ahead, class tables replaced as the drift moves,
a step and a counts() call thrown in now and then), every stream mirrored by the oracle call by call: per-call counts of
every run, the samples of every 64th run, the final states bit for bit (~100 M frames per stream).

--hours H: the streams are aged to H hours of audio (24 h = 3.8 G frames per stream at 44.1 kHz,
29 k runs of 256 calls) through rsmp_fir_lockstep_run alone; the oracle follows by orc_fir_skip_calls (the reference's
f64 position recurrence output by output, no samples) and, at every checkpoint, the batch's states must equal the
oracle's bit for bit, a fully mirrored run (counts of all 256 calls, samples within 1e-6 RMS) follows, no call of it may
have left the planner's fast path (run_slow_calls() == 0) and no stream may carry a status flag (4 = its drift left the
class tables' tolerance: the reference-form cliff of rounds 1-4 at ~8 hours)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import resampler_amd as ra
from oracle import pyoracle as o
from resampler_amd import synth

args = sys.argv[1:]
hours = None
if args and args[0] == "--hours":
    hours = float(args[1])
    args = args[2:]
frames = 512
dev = torch.device("cuda:0")
kind = o.CONVOLVE_AVX_FMA if o.have_avx_fma() else o.CONVOLVE_SCALAR


def make(pairs, k):
    hs = [ra.ResamplerFir.new_from_hz(2, i, o_, ra.Latency.Sample64, ra.Attenuation.Db90) for i, o_ in pairs]
    refs = [o.OracleFir(2, i, o_, 128, 90, kind) for i, o_ in pairs]
    x = synth.fast_noise(2 * frames * (k + 1), seed=77)
    d_in = [torch.from_numpy(x).to(dev) for _ in hs]
    caps = [h.buffer_size_output() for h in hs]
    d_out = [torch.zeros((k + 1) * c, device=dev) for c in caps]
    ls = ra.FirLockstep(hs, frames)
    ls.bind_caps(d_in, d_out, caps)
    return hs, refs, x, d_out, ls


if hours is not None:
    k = 256
    pairs = [(44100, 48000), (48000, 44100), (96000, 44100), (44100, 96000), (48000, 96000), (96000, 48000)]
    hs, refs, x, d_out, ls = make(pairs, k)
    xr = x[:2 * frames * k]                       # the span every run reads
    total_runs = int(np.ceil(hours * 3600 * 44100 / (k * frames)))
    every = max(1, total_runs // 12)              # a dozen checkpoints
    t0 = time.time()
    done = since = 0
    worst = 0.0
    flags = 0
    while done < total_runs:
        n = min(every, total_runs - done)
        for _ in range(n - 1):
            ls.run(k, frames, 0, append=False)
        done += n - 1
        since += n - 1
        # the oracle catches up: control flow only, then the buffered frames (the tail of the span) back into its ring
        ls.sync()
        for i, (h, r) in enumerate(zip(hs, refs)):
            r.skip_calls(since * k, frames)
            assert h.state() == r.state(), (done, i, h.state(), r.state())
            r.seek(r.state(), xr)
        since = 0
        # one run mirrored call by call
        ls.run(k, frames, 0, append=False)
        done += 1
        cons, prod = ls.run_counts()
        slow = ls.run_slow_calls()
        for i, r in enumerate(refs):
            y, calls = r.resample_all(xr, 2 * frames, max_calls=k + 4)
            assert calls.shape[0] == k and (calls[:, 0] == cons[:, i]).all() and (calls[:, 1] == prod[:, i]).all(), (done, i)
            g = d_out[i][:y.size].cpu().numpy()
            worst = max(worst, float(np.sqrt(np.mean((g.astype(np.float64) - y) ** 2))))
        st = ls.status()
        flags |= int(np.bitwise_or.reduce(st))
        audio_h = done * k * frames / 44100 / 3600
        print(f"run {done:6d}  {audio_h:6.2f} h of 44.1 kHz audio per stream: states equal, run mirrored (worst RMS so far {worst:.2e}), "
              f"slow calls {slow}, status {int(st.max())}, drift classes {ls.stats()['drift_classes']}, "
              f"table replacements {ls.table_rebinds()}, {time.time() - t0:.0f} s", flush=True)
        assert worst <= 1e-6, worst
        assert slow == 0, slow
        assert int(st.max()) == 0, st
    ls.sync()
    for h, r in zip(hs, refs):
        assert h.state() == r.state()
    print(f"{done} runs of {k} calls x {len(hs)} streams = {done * k * frames / 1e9:.2f} G frames per stream "
          f"({done * k * frames / 44100 / 3600:.1f} h at 44.1 kHz, {done * k * frames / 96000 / 3600:.1f} h at 96 kHz): every checkpoint's "
          f"states equal the reference recurrence's bit for bit, worst RMS of the mirrored runs {worst:.2e}, no slow call, "
          f"no status flag, {ls.table_rebinds()} table replacements, stats {ls.stats()}, {time.time() - t0:.0f} s")
    sys.exit(0)

runs = int(args[0]) if len(args) > 0 else 3000
k = int(args[1]) if len(args) > 1 else 64
pairs = [(44100, 48000), (48000, 44100), (96000, 44100), (44100, 96000)]
hs, refs, x, d_out, ls = make(pairs, k)
orr = [np.zeros(r.buffer_size_output(), np.float32) for r in refs]
t0 = time.time()
worst = 0.0
for p in range(runs):
    ls.run(k, frames, 0, append=False)
    check = p % 64 == 63 or p == runs - 1
    extra = p % 97 == 96
    cons, prod = ls.run_counts()
    if extra:
        ls.step(frames, k * frames, append=True)
        c1, p1 = ls.counts()
    for i, r in enumerate(refs):
        want = []
        for s in range(k + (1 if extra else 0)):
            rc, cr, pr = r.resample(x[s * frames * 2:(s + 1) * frames * 2], orr[i])
            cg, pg = (cons[s][i], prod[s][i]) if s < k else (c1[i], p1[i])
            assert rc == 0 and (int(cg), int(pg)) == (cr, pr), (p, s, i, (cg, pg), (cr, pr))
            if check:
                want.append(orr[i][:pr].copy())
        if check:
            w = np.concatenate(want)
            g = d_out[i][:w.size].cpu().numpy()
            e = float(np.sqrt(np.mean((g.astype(np.float64) - w) ** 2)))
            worst = max(worst, e)
            assert e <= 1e-6, (p, i, e)
assert int(ls.status().max()) == 0, ls.status()
ls.sync()
for h, r in zip(hs, refs):
    assert h.state() == r.state()
print(f"{runs} runs of {k} calls x {len(hs)} streams ({runs * k * frames / 1e6:.0f} M frames per stream): counts of every call equal, "
      f"worst RMS of the checked runs {worst:.2e}, final states equal, {ls.table_rebinds()} table replacements, {time.time() - t0:.0f} s")
