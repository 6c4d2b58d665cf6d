"""tools/soak_lockstep.py [runs] [k] -- a long lock-step session on one small mixed batch: thousands of runs of k calls
(the next one planned ahead, class tables replaced as the drift moves, a step and a counts() call thrown in now and
then), every stream mirrored by the oracle: per-call counts of every run, the samples of every 64th run, the final states
bit for bit.  No hang, no status flag, parity at the end of ~100 M frames per stream."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import resampler_amd as ra
from oracle import pyoracle as o
from resampler_amd import synth
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 64
frames = 512
dev = torch.device("cuda:0")
pairs = [(44100, 48000), (48000, 44100), (96000, 44100), (44100, 96000)]
hs = [ra.ResamplerFir.new_from_hz(2, i, o_, ra.Latency.Sample64, ra.Attenuation.Db90) for i, o_ in pairs]
kind = o.CONVOLVE_AVX_FMA if o.have_avx_fma() else o.CONVOLVE_SCALAR
refs = [o.OracleFir(2, i, o_, 128, 90, kind) for i, o_ in pairs]
x = synth.fast_noise(2 * frames * (k + 1), seed=77)
d_in = [torch.from_numpy(x).to(dev) for _ in hs]
caps = [h.buffer_size_output() for h in hs]
d_out = [torch.zeros((k + 1) * c, device=dev) for c in caps]
ls = ra.FirLockstep(hs, frames)
ls.bind_caps(d_in, d_out, caps)
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
