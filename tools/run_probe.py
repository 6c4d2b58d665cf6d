"""tools/run_probe.py [streams] [k] -- rsmp_fir_lockstep_run on config 4's batch: wall clock per run and the calls the
planner's fast path declined."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import resampler_amd as ra
from resampler_amd import sharding, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
own = torch.cuda.Stream() if os.environ.get("PROBE_STREAM") else None   # (a stream of the caller's own instead of the null stream)
if own and os.environ.get("PROBE_SETSTREAM"):
    torch.cuda.set_stream(own)
specs = sharding.mixed_rate_batch(n, 2, 512)
hs = [ra.ResamplerFir.new_from_hz(2, s.in_hz, s.out_hz, ra.Latency.Sample64, ra.Attenuation.Db90) for s in specs]
caps = [h.buffer_size_output() for h in hs]
x = torch.from_numpy(synth.fast_noise(k * 512 * 2, seed=1)).to(dev)
gains = torch.linspace(0.5, 1.0, n, device=dev) if os.environ.get("PROBE_GAINS") else torch.ones(n, device=dev)
d_in = [(x * gains[i]).contiguous() for i in range(n)]
d_out = [torch.empty(c + k * 2 * (512 * s.out_hz // s.in_hz + 2), device=dev) for c, s in zip(caps, specs)]
ls = ra.FirLockstep(hs, 512)
ls.bind_caps(d_in, d_out, caps)
sp = own.cuda_stream if own else None
REPS = int(os.environ.get('PROBE_REPS', '4'))
RUNS = int(os.environ.get('PROBE_RUNS', '8'))
for rep in range(REPS):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(RUNS):
        ls.run(k, 512, 0, append=False, stream=sp)
    th = (time.perf_counter() - t0) / RUNS   # (the host's share: eight runs enqueued, nothing waited for)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / RUNS
    print(f"streams {n} k {k}: {dt * 1e6:.1f} us per run ({th * 1e6:.1f} us of host calls), {dt * 1e6 / k:.2f} us per step, slow calls {ls.run_slow_calls()} of {n * k}")
