"""tools/run_bulk_probe.py [streams] [frames] -- GPU box: a bulk batch in distinct states through rsmp_fir_lockstep_run_bulk (planned on
the device), launch after launch on the streams' own state: wall clock per launch; under `tools/kt_probe.sh` the kernels it is made of."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import resampler_amd as ra
from resampler_amd import synth
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
frames_call = 256
dev = torch.device("cuda:0")
hs = [ra.ResamplerFir.new(2, ra.SampleRate.Hz44100, ra.SampleRate.Hz48000, ra.Latency.Sample64, ra.Attenuation.Db90) for _ in range(S)]
warm = np.zeros(2 * 4096, np.float32)
if not os.environ.get('PROBE_NOWARM'):
    for i, h in enumerate(hs):
        h.resample_bulk(warm[: 2 * (64 + 37 * i)], 512)
base = torch.from_numpy(synth.sweep(N, 2, 44100.0)).to(dev)
d_in = [base.clone() for _ in range(S)]
caps = [h.buffer_size_output() for h in hs]   # (the per-CALL capacity, as the reference sizes a call's buffer)
bound = [h.bulk_output_bound(2 * N, 2 * frames_call) for h in hs]
d_out = [torch.empty(c, device=dev) for c in bound]
if os.environ.get('PROBE_PRELAUNCH'):
    b = ra.FirBatch(hs); b.bind(d_in, d_out); b.resample_bulk_device(512, ra.torch_stream()); torch.cuda.synchronize()
ls = ra.FirLockstep(hs, frames_call)
ls.bind_caps(d_in, d_out, caps)
ts = torch.cuda.Stream()
st = ts.cuda_stream if not os.environ.get('PROBE_OWN_STREAM') else None   # (a stream of the caller's, as bench.py has; None: the batch's own)
if os.environ.get('PROBE_LEGACY'):
    st = ra.STREAM_LEGACY
times = []
for rep in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ls.run_bulk(N, frames_call, 0, append=False, stream=st)
    torch.cuda.synchronize(); times.append((time.perf_counter() - t0) * 1e3)
print("streams %d frames %d, device-planned bulk launches (ms): %s" % (S, N, " ".join("%.2f" % t for t in times)))
