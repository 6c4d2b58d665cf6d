import os, sys, subprocess, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import resampler_amd as ra
    from resampler_amd import synth
    f = ra.ResamplerFft.new(2, ra.SampleRate.Hz44100, ra.SampleRate.Hz48000)
    n_in, n_out = f.chunk_size_input(), f.chunk_size_output()
    x = synth.sweep(3 * n_in // 2, 2, 44100.0)
    outs = []
    o = np.zeros(n_out, np.float32)
    for b in range(3):
        f.resample(x[b * n_in:(b + 1) * n_in], o)
        outs.append(o.copy())
    np.save(sys.argv[2], np.concatenate(outs))
else:
    for mode, path in (("0", "/tmp/old.npy"), ("1", "/tmp/new.npy")):
        env = dict(os.environ, RSMP_DEBUG="1", RSMP_FFT_WAVE=mode)
        subprocess.check_call([sys.executable, __file__, "child", path], env=env)
    a, b = np.load("/tmp/old.npy"), np.load("/tmp/new.npy")
    d = np.abs(a.astype(np.float64) - b)
    print("max", d.max(), "rms", np.sqrt((d ** 2).mean()), "nonzero", (d > 0).sum(), "of", d.size)
    for blk in range(3):
        seg = d[blk * 2560:(blk + 1) * 2560]
        bad = np.flatnonzero(seg > 1e-6)
        print("block", blk, "bad", bad.size, bad[:16], "ch0 bad", (bad % 2 == 0).sum(), "ch1 bad", (bad % 2 == 1).sum())
        if bad.size:
            print("  old", a[blk * 2560 + bad[:6]], "new", b[blk * 2560 + bad[:6]])
