#!/usr/bin/env python3
"""bench.py -- headline benchmark: Msamples/s (input f32 values) of the 128-tap polyphase FIR,
2 ch 44.1 kHz -> 48 kHz (BASELINE.json configs[1]), inputs resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W]            headline (FIR, weak scaling)
  python bench.py --config c4 [--gpus N]                         BASELINE config 4 (strong scaling)
  python bench.py --path fft                                     ResamplerFft line alone

A step = one pass of the hot path over one batch: `--streams` independent 2-channel streams, each
a fresh 2^20-frame sine sweep (reset + the reference's bulk driver loop with 512-value chunks,
resample/src/main.rs:226-254), all streams in ONE launch of the periodic FIR kernel.  One stream
alone is 8 MiB in / 8.7 MiB out -- microseconds of HBM time -- so the single-GPU workload is a
batch of them (weak scaling: every rank owns `--streams` streams; no data-path collective).

--gpus N > 1 without a launcher spawns N fresh rank processes (one per GPU, RCCL group over
127.0.0.1) BEFORE anything touches the GPU and prints rank 0's line; under
`python -m torch.distributed.run` (RANK / WORLD_SIZE in the environment) the process is one rank.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel, timed with HIP events on the
launch stream inside the library (rsmp_fir_set_profiling); `cpu_baseline` is the oracle's AVX+FMA
restatement of the reference path on the host cores; `secondary` (N = 1 only) carries the FFT
path, the exact-f32 and vector FIR kernels, the cold-plan / distinct-state FIR step and config 4.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
IN_HZ, OUT_HZ, CHANNELS = 44100, 48000, 2
FIR_DTYPE = {4: "f32 (bf16x3-split products, f32 accumulate)", 5: "f32 (fp16x2-split products, f32 accumulate)"}
KERNEL_NAMES = {0: "fir_generic_kernel", 1: "fir_periodic_kernel (vector)", 2: "fir_periodic_db_kernel (vector)",
                3: "fir_periodic_db_kernel (exact-f32 MFMA)", 4: "fir_split_kernel (bf16x3 MFMA)",
                5: "fir_split_kernel (fp16x2 MFMA)"}


# ------------------------------------------------------------------------------------------------
# process layout
# ------------------------------------------------------------------------------------------------
def spawn_ranks(args) -> int:
    """--gpus N without a launcher: start N rank processes of this script (one per GPU) before this
    process has initialised HIP, wait for them, and pass rank 0's JSON line through."""
    import torch
    have = torch.cuda.device_count()      # does not initialise the GPU on this image
    if have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        p.wait()
        rc = rc or p.returncode
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return rc


class Ctx:
    """One rank: device, process group, barrier, max-over-ranks timing."""

    def __init__(self, args):
        import torch
        self.torch = torch
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != max(1, args.gpus) and self.rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}; using WORLD_SIZE", file=sys.stderr)
        self.dist = None
        if self.world > 1 or getattr(args, "feed", "resident") == "rccl":
            # (--feed rccl on one GPU: a world of one rank, so that RCCL's init, group launch and the ordering of
            # its work against the resampling kernels run on whatever hardware there is)
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.world == 1 and "MASTER_PORT" not in os.environ:
                with socket.socket() as so:
                    so.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(so.getsockname()[1])
            dist.init_process_group(backend="nccl", rank=self.rank, world_size=self.world,
                                    device_id=torch.device(f"cuda:{self.local_rank}"))
            self.dist = dist
        import resampler_amd as ra
        if not torch.cuda.is_available() or ra.device_count() <= self.local_rank:
            raise SystemExit("bench.py needs a HIP device per rank (no CPU fallback)")
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device(f"cuda:{self.local_rank}")
        # Everything this rank enqueues -- torch fills and copies, RCCL work (its handles block the CURRENT torch
        # stream) and the library's launches -- goes to ONE non-default stream, so a step's stages are ordered by
        # the stream itself.  (torch's default stream has handle 0, which the C ABI reads as "the handle's own
        # non-blocking stream": launches there would be ordered against nothing torch does.)
        self.torch_stream = torch.cuda.Stream(device=self.dev)
        torch.cuda.set_stream(self.torch_stream)
        self.stream = ra.torch_stream(self.torch_stream)

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.dist:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, dt: float) -> float:
        if not self.dist:
            return dt
        t = self.torch.tensor([dt], device=self.dev, dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, v: float) -> float:
        if not self.dist:
            return v
        t = self.torch.tensor([v], device=self.dev, dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.dist:
            self.dist.barrier()
            self.dist.destroy_process_group()


def timed(ctx: Ctx, step, steps: int, warmup: int):
    """W untimed steps, then exactly K steps bracketed by barrier + synchronize; max over ranks."""
    for _ in range(warmup):
        step()
    ctx.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    host_dt = time.perf_counter() - t0
    ctx.barrier()
    dt = time.perf_counter() - t0
    return ctx.max_over_ranks(dt), host_dt


def device_copy_gbs(ctx: Ctx, nbytes: int) -> float:
    """What a plain device-to-device copy of `nbytes` (read) + `nbytes` (written) reaches on this GPU, in GB/s of
    read + written bytes: the PRACTICAL ceiling of a kernel that streams as much in as out, next to the 8 TB/s
    the roofline is priced against ('also report a measured device-copy GB/s as the practical
    peak').  The library's own 16-bytes-per-lane copy kernel (rsmp_device_stream_copy; round 3 timed torch's
    Tensor.copy_, which stops at 4.8-5.4 TB/s) on the bench's stream, torch events, 20 copies after 5."""
    import resampler_amd as ra
    torch = ctx.torch
    if not hasattr(ra.lib(), "rsmp_device_stream_copy"):   # (an A/B library of an older commit)
        return 0.0
    n = max(1 << 20, int(nbytes) // 4) // 4 * 4
    src = torch.empty(n, device=ctx.dev, dtype=torch.float32).normal_()
    dst = torch.empty_like(src)
    for _ in range(5):
        ra.device_stream_copy(src, dst, ctx.stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ra.device_stream_copy(src, dst, ctx.stream)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / 20
    del src, dst
    return 2.0 * 4.0 * n / (ms * 1e-3) / 1e9


def spinup(ctx: Ctx, step, seconds: float) -> None:
    """Untimed steps of the same workload until `seconds` have passed, before the contract's warmup
    steps.  Measured on the pool's MI355X: the first ~100 launches after idle run ~18 % slower than
    the steady state -- a 20-step run never leaves the governor's ramp, so without this the line
    reports the ramp, not the kernel.  The timed region is unchanged."""
    if seconds <= 0:
        return
    t0 = time.perf_counter()
    while True:
        for _ in range(20):
            step()
        ctx.torch.cuda.synchronize()
        done = time.perf_counter() - t0 >= seconds
        if ctx.dist:   # every rank runs the same number of steps (a step may contain collectives: --feed rccl)
            done = ctx.max_over_ranks(1.0 if done else 0.0) > 0.0
        if done:
            break


# ------------------------------------------------------------------------------------------------
# CPU baselines (oracle = port of the reference; test infrastructure, used here only as `cpu_baseline`)
# ------------------------------------------------------------------------------------------------
def cpu_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return model, cores


def cores_granted():
    """What the box grants this process: the cgroup's CPU quota in cores (cpu.max: quota / period; None = no limit or not
    readable).  `host_cores` (the affinity mask) is what the machine has, not what a container may use of it -- the pool's
    boxes report 256 and grant about a dozen."""
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt[0] == "max":
                    return None
                return round(int(txt[0]) / int(txt[1]), 2)
            q = int(txt[0])
            if q <= 0:
                return None
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                return round(q / int(f.read().split()[0]), 2)
        except (OSError, ValueError, IndexError):
            continue
    return None


def _fir_cpu_worker(x, seconds, chunk, out, idx, specs=(CHANNELS, IN_HZ, OUT_HZ, 128, 90), kind=None):
    from oracle import pyoracle as orc
    if kind is None:
        kind = orc.CONVOLVE_AVX_FMA if orc.have_avx_fma() else orc.CONVOLVE_SCALAR
    r = orc.OracleFir(specs[0], specs[1], specs[2], specs[3], specs[4], kind)
    y = np.empty(int(x.size / r.ratio) + 4 * r.buffer_size_output(), np.float32)
    r.resample_all_into(x[: 2 * 65536], chunk, y)  # warm caches / page in
    y[:] = 0
    t0 = time.perf_counter()
    values = 0
    while True:
        r.reset()
        r.resample_all_into(x, chunk, y)           # the CLI's driver loop in C (ctypes releases the GIL for the call)
        values += x.size
        if time.perf_counter() - t0 >= seconds:
            break
    out[idx] = (values, time.perf_counter() - t0)


def _fft_cpu_worker(x, seconds, out, idx, simd=True):
    from oracle import pyoracle as orc
    r = orc.OracleFft(CHANNELS, IN_HZ, OUT_HZ, simd=simd)
    y = np.zeros((x.size // r.chunk_size_input() + 1) * r.chunk_size_output(), np.float32)
    r.resample_all_into(x[: 8 * r.chunk_size_input()], y)
    t0 = time.perf_counter()
    values = 0
    while True:
        r.resample_all_into(x, y)                  # the CLI's driver loop in C (one ctypes call per pass)
        values += x.size
        if time.perf_counter() - t0 >= seconds:
            break
    out[idx] = (values, time.perf_counter() - t0)


def _all_cores(worker, args, seconds, cores):
    n = min(cores, 256)
    res = [None] * n
    ths = [threading.Thread(target=worker, args=args + (seconds, res, i)) for i in range(n)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    wall = time.perf_counter() - t0
    return round(sum(r[0] for r in res) / wall / 1e6, 1), n, wall


# BASELINE.md section 1, converted to this metric (input values/s, one core of a Ryzen 9 9950X3D, v0.3.3 criterion
# benches over four rate pairs; the changelog does not say which end belongs to 44.1 -> 48 kHz)
PUBLISHED_REF = {"fir": {"Msamples_in_per_s": [120, 130], "source": "CHANGELOG.md:77 (503-540 MiB/s out)", "cpu": "Ryzen 9 9950X3D, 1 thread"},
                 "fft": {"Msamples_in_per_s": [190, 290], "source": "CHANGELOG.md:76 (780-1192 MiB/s out)", "cpu": "Ryzen 9 9950X3D, 1 thread"}}


def cpu_baseline_fir(frames: int, seconds: float, all_cores_seconds: float):
    """Oracle (port of the reference AVX+FMA path, fir/avx.rs + resampler_fir.rs), built as BASELINE.md section 2
    prescribes (oracle/liboracle_native.so: -O3 -mavx2 -mfma; bit-identical to the portable build,
    tests/test_oracle_native.py): the same 2 ch 44.1k->48k 128-tap sweep in 512-value calls, repeated for ~`seconds`
    on ONE core; then one stream per core on all host cores (the reference is single-threaded per instance)."""
    from oracle import pyoracle as orc
    from resampler_amd import synth
    model, cores = cpu_info()
    native = orc.use_native(True)
    try:
        x = synth.sweep(frames, CHANNELS, float(IN_HZ))
        res = [None]
        _fir_cpu_worker(x, seconds, 512, res, 0)
        one = res[0][0] / res[0][1] / 1e6
        line = {
            "value": round(one, 3), "unit": "Msamples/s", "cores": 1, "kind": "port", "model": model,
            "host_cores": cores, "build_flags": orc.build_flags(), "published_ref": PUBLISHED_REF["fir"],
            "sample": f"{res[0][0] // x.size} passes of one {frames}-frame 2ch sweep, 512-value calls, "
                      f"{'AVX+FMA' if orc.have_avx_fma() else 'scalar'} convolve, {res[0][1]:.1f} s",
        }
        granted = cores_granted()
        if granted is not None:
            line["cgroup_cpu_quota_cores"] = granted
        # The reference's own measurement convention, so that the port can be laid beside CHANGELOG.md:77 (503-540 MiB/s):
        # benches/benchmark_resampler_fir.rs -- two channels, Sample64 / Db90, ONE 1024-value block of LCG white noise fed to
        # resample() over and over, throughput counted as round(1024 * out / in) OUTPUT values x 4 bytes per call, four rate
        # pairs -- through both leaves: AVX+FMA (north_star's baseline) and AVX-512, which the reference's runtime dispatch
        # takes FIRST on a CPU that has it (resampler_fir.rs:331-345) and the published Zen 5 figures therefore ran.
        from resampler_amd import synth as _synth
        noise = _synth.lcg_noise(1024)
        kinds = [("avx_fma", orc.CONVOLVE_AVX_FMA)] if orc.have_avx_fma() else [("scalar", orc.CONVOLVE_SCALAR)]
        if orc.have_avx512f():
            kinds.append(("avx512", orc.CONVOLVE_AVX512))
        conv = {}
        for name, kind in kinds:
            per_pair = {}
            for a, b in ((48000, 96000), (22050, 48000), (44100, 48000), (48000, 44100)):
                r = orc.OracleFir(2, a, b, 128, 90, kind)
                out = np.zeros(r.buffer_size_output(), np.float32)
                r.bench_calls(noise, out, 2000)
                t0 = time.perf_counter()
                calls = 0
                while time.perf_counter() - t0 < 0.35:
                    r.bench_calls(noise, out, 4000)
                    calls += 4000
                dt = time.perf_counter() - t0
                per_pair[f"{a}->{b}"] = round(calls * round(1024 * b / a) * 4 / dt / 2**20, 1)
            conv[name] = per_pair
        line["ref_convention_MiBps_out"] = conv
        if orc.have_avx512f():
            res512 = [None]
            _fir_cpu_worker(x, min(seconds, 3.0), 512, res512, 0, kind=orc.CONVOLVE_AVX512)
            line["avx512"] = {"value": round(res512[0][0] / res512[0][1] / 1e6, 3), "unit": "Msamples/s", "cores": 1,
                              "what": "the same workload through fir/avx512.rs's leaf (oracle/fir.c: orc_convolve_interp_avx512) -- "
                                      "the path the reference's dispatch takes first on this CPU and on the Zen 5 of its changelog"}
        best = kinds[-1][0]
        lo, hi = min(conv[best].values()), max(conv[best].values())
        base = kinds[0][0]
        crit_in = conv[base]["44100->48000"] * 2**20 / (round(1024 * 48000 / 44100) * 4) * 1024 / 1e6   # the criterion loop, as input values/s
        line["criterion_loop_Msamples_in_per_s"] = round(crit_in, 1)
        line["gap_note"] = (f"published 503-540 MiB/s out (= 120-130 M in/s) is the criterion convention on a Ryzen 9 9950X3D (Zen 5, 5.7 GHz), "
                            f"whose runtime dispatch takes the AVX-512 leaf (resampler_fir.rs:331-345).  This host in that convention: "
                            f"{lo}-{hi} MiB/s with the {best} leaf, {min(conv[base].values())}-{max(conv[base].values())} with the {base} leaf "
                            f"({conv[base]['44100->48000']} at 44.1->48 kHz = {crit_in:.0f} M in/s, against `value` {one:.0f} M in/s for the CLI loop "
                            f"over the 2^20-frame sweep with the same leaf).  So of the distance to the published band, "
                            f"x{(conv[best]['44100->48000'] / conv[base]['44100->48000']):.2f} is the leaf north_star names (AVX+FMA, not the one the "
                            f"reference would pick here), x{(crit_in / one if one > 0 else 0):.2f} the bench's hot 1024-value block against the CLI loop, "
                            f"the rest ({(503.0 / hi):.2f}-{(540.0 / lo):.2f}x) clock and core (see `model`)")
        if all_cores_seconds > 0 and cores > 1:
            xs = x[: 2 * (1 << 18)]                    # a quarter of the sweep per thread: bounded sample
            v, n, wall = _all_cores(lambda x_, c_, s_, r_, i_: _fir_cpu_worker(x_, s_, c_, r_, i_), (xs, 512), all_cores_seconds, cores)
            # (cores_effective: what the threads got between them, in units of the one-thread run above -- the affinity mask
            # says 256 where the container is granted about twelve)
            line["all_cores"] = {"value": v, "unit": "Msamples/s", "threads": n, "cores": n,
                                 "cores_effective": round(v / one, 1) if one > 0 else None,
                                 "sample": f"one stream per thread, 2^18-frame sweeps, {wall:.1f} s"}
            line["cores_effective"] = line["all_cores"]["cores_effective"]
    finally:
        orc.use_native(False)
    if not native:
        line["note"] = "this CPU lacks AVX2+FMA: portable build timed"
    return line


def cpu_baseline_fft(seconds: float, all_cores_seconds: float = 0.0):
    """OracleFft (port of resampler_fft.rs:182-240 + src/fft) through the reference's AVX + FMA code path -- its
    butterflies 3 / 4 / 5 / 7 / 8 and real <-> complex passes restated intrinsic for intrinsic (oracle/fft_avx.c; packed
    twiddles as RadixFFT::new) --, native build, the CLI's driver loop over 256 blocks on one core, then one stream per
    core; the scalar-butterfly figure of rounds 1-3 rides along as `scalar_value`."""
    from oracle import pyoracle as orc
    from resampler_amd import synth
    model, cores = cpu_info()
    orc.use_native(True)
    try:
        n_in = orc.OracleFft(CHANNELS, IN_HZ, OUT_HZ).chunk_size_input()
        blocks = 256
        x = synth.sweep(blocks * n_in // CHANNELS, CHANNELS, float(IN_HZ))
        res = [None, None]
        simd = orc.have_avx_fma()
        _fft_cpu_worker(x, seconds, res, 0, simd)
        _fft_cpu_worker(x, min(seconds, 2.0), res, 1, False)
        values, dt = res[0]
        line = {"value": round(values / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port-avx" if simd else "port",
                "model": model, "host_cores": cores, "build_flags": orc.build_flags(), "published_ref": PUBLISHED_REF["fft"],
                "scalar_value": round(res[1][0] / res[1][1] / 1e6, 3),
                "sample": f"{values // n_in} blocks of 1176 frames, " + ("the reference's AVX + FMA butterflies and real <-> complex "
                          "passes (oracle/fft_avx.c)" if simd else "scalar butterflies (no AVX + FMA on this CPU)") + f", {dt:.1f} s"}
        granted = cores_granted()
        if granted is not None:
            line["cgroup_cpu_quota_cores"] = granted
        if all_cores_seconds > 0 and cores > 1:
            v, n, wall = _all_cores(lambda x_, s_, r_, i_: _fft_cpu_worker(x_, s_, r_, i_), (x,), all_cores_seconds, cores)
            one = values / dt / 1e6
            line["all_cores"] = {"value": v, "unit": "Msamples/s", "threads": n, "cores": n,
                                 "cores_effective": round(v / one, 1) if one > 0 else None,
                                 "sample": f"one stream per thread, 256-block passes, {wall:.1f} s"}
            line["cores_effective"] = line["all_cores"]["cores_effective"]
    finally:
        orc.use_native(False)
    return line


# ------------------------------------------------------------------------------------------------
# GPU workloads
# ------------------------------------------------------------------------------------------------
def traffic_from_profiles(name: str, kernel_name=None):
    """HBM bytes per launch of the workload's dominant kernel from the committed PMC passes
    (profiles/traffic_latest.json, written by tools/profile_round.sh on the GPU box: separate FETCH_SIZE /
    WRITE_SIZE passes with the guide's gfx950 corrections; this run does not collect counters itself)."""
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        ent = json.load(open(tpath)).get(name)
    except Exception:
        return None
    if not isinstance(ent, dict):
        return None
    if kernel_name is not None and ent.get("kernel") not in (None, kernel_name):
        return None   # the counters were collected for another kernel
    # ... or for other SOURCES of it: the profile names the kernel sources it was taken with, and a kernel changed since
    # reports no traffic rather than the old one (tools/profile_round.sh renews it)
    from resampler_amd import provenance
    if ent.get("kernel_sources_sha") != provenance.kernel_sources_sha(name):
        return None
    return ent.get("hbm_bytes_per_launch")


def make_fir_batch(ctx: Ctx, ra, S: int, N: int, chunk: int, kernel, distinct_states: bool = False):
    from resampler_amd import synth
    torch = ctx.torch
    handles = []
    for _ in range(S):
        h = ra.ResamplerFir.new(CHANNELS, ra.SampleRate.Hz44100, ra.SampleRate.Hz48000,
                                ra.Latency.Sample64, ra.Attenuation.Db90, device=ctx.local_rank)
        h.set_kernel(kernel)
        handles.append(h)
    if distinct_states:   # every stream has already run for a different time
        warm = np.zeros(CHANNELS * 4096, np.float32)
        for i, h in enumerate(handles):
            h.resample_bulk(warm[: CHANNELS * (64 + 37 * i)], chunk)
    base = torch.from_numpy(synth.sweep(N, CHANNELS, float(IN_HZ))).to(ctx.dev)
    gains = torch.linspace(0.5, 1.0, S, device=ctx.dev)
    d_in = [(base * gains[i]).contiguous() for i in range(S)]
    cap = max(h.bulk_output_bound(CHANNELS * N, chunk) for h in handles)
    d_out = [torch.empty(cap, device=ctx.dev, dtype=torch.float32) for _ in range(S)]
    batch = ra.FirBatch(handles)
    batch.bind(d_in, d_out)
    return handles, batch


def fir_kernel_point(ctx: Ctx, ra, args, kernel, steps: int):
    """Kernel time / roofline fraction of one FIR kernel flavour on the headline workload."""
    S, N = args.streams, args.frames
    handles, batch = make_fir_batch(ctx, ra, S, N, args.chunk, kernel)

    def step():
        batch.reset()
        return batch.resample_bulk_device(args.chunk, ctx.stream)
    consumed, produced = step()
    out_values = int(sum(produced))
    spinup(ctx, step, 0.5)
    handles[0].set_profiling(True)
    for _ in range(steps):
        step()
    k_ms, _ = handles[0].mean_kernel_ms()
    handles[0].set_profiling(False)
    alg = 4.0 * (S * CHANNELS * N + out_values) + 4.0 * 1024 * 128
    ach = alg / (k_ms * 1e-3) / 1e9
    return {"kernel": KERNEL_NAMES.get(handles[0].kernel_variant(), "?"), "kernel_ms": round(k_ms, 4),
            "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4),
            "Msamples_in_per_s_kernel": round(S * CHANNELS * N / (k_ms * 1e-3) / 1e6, 1)}


def fir_channels_point(ctx: Ctx, ra, args, channels: int, steps: int):
    """Kernel time / roofline fraction of the default FIR kernel for another channel count: the headline's rate pair
    and taps, 128 / channels streams x `--frames` frames per launch (the split matrix kernel on channel pairs)."""
    from resampler_amd import synth
    torch = ctx.torch
    S, N = max(1, 128 // channels), args.frames
    handles = [ra.ResamplerFir.new(channels, ra.SampleRate.Hz44100, ra.SampleRate.Hz48000, ra.Latency.Sample64,
                                   ra.Attenuation.Db90, device=ctx.local_rank) for _ in range(S)]
    x = torch.from_numpy(synth.fast_noise(N * channels, seed=7)).to(ctx.dev)
    gains = torch.linspace(0.5, 1.0, S, device=ctx.dev)
    d_in = [(x * gains[i]).contiguous() for i in range(S)]   # a buffer of its own per stream: every byte comes from HBM
    chunk = 512 * channels
    cap = handles[0].bulk_output_bound(channels * N, chunk)
    d_out = [torch.empty(cap, device=ctx.dev, dtype=torch.float32) for _ in range(S)]
    batch = ra.FirBatch(handles)
    batch.bind(d_in, d_out)

    def step():
        batch.reset()
        return batch.resample_bulk_device(chunk, ctx.stream)
    consumed, produced = step()
    out_values = int(sum(produced))
    spinup(ctx, step, 0.3)
    handles[0].set_profiling(True)
    for _ in range(steps):
        step()
    k_ms, _ = handles[0].mean_kernel_ms()
    handles[0].set_profiling(False)
    # ... and the wall clock of a step (reset + launch + the repair launch behind it, which the kernel's events do not bracket:
    # round 5's three-channel launch spent 1.0 ms THERE -- profiles/r06/odd_channels_repair.txt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        step()
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) / 8 * 1e3
    alg = 4.0 * (S * channels * N + out_values) + 4.0 * 1024 * 128
    ach = alg / (k_ms * 1e-3) / 1e9
    point = {"kernel": KERNEL_NAMES.get(handles[0].kernel_variant(), "?"), "streams": S, "kernel_ms": round(k_ms, 4),
             "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4),
             "step_ms_wall": round(wall_ms, 4), "frac_wall": round(alg / (wall_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
             "Msamples_in_per_s_kernel": round(S * channels * N / (k_ms * 1e-3) / 1e6, 1)}
    del batch, handles
    return point


def fir_nonperiodic_point(ctx: Ctx, ra, args):
    """A ratio without a short period (ResamplerFir::new_from_hz with arbitrary rates: 2 ch 44100 -> 47999 Hz), the headline's batch
    shape: fir_generic_bulk.hip (tiles sorted by phase row).  Wall clock of a step."""
    from resampler_amd import synth
    torch = ctx.torch
    S, N = args.streams, args.frames
    handles = [ra.ResamplerFir.new_from_hz(CHANNELS, 44100, 47999, ra.Latency.Sample64, ra.Attenuation.Db90, device=ctx.local_rank) for _ in range(S)]
    x = torch.from_numpy(synth.fast_noise(N * CHANNELS, seed=11)).to(ctx.dev)
    gains = torch.linspace(0.5, 1.0, S, device=ctx.dev)
    d_in = [(x * gains[i]).contiguous() for i in range(S)]
    cap = handles[0].bulk_output_bound(CHANNELS * N, args.chunk)
    d_out = [torch.empty(cap, device=ctx.dev, dtype=torch.float32) for _ in range(S)]
    batch = ra.FirBatch(handles)
    batch.bind(d_in, d_out)

    def step():
        batch.reset()
        return batch.resample_bulk_device(args.chunk, ctx.stream)
    consumed, produced = step()
    out_values = int(sum(produced))
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) / 5 * 1e3
    alg = 4.0 * (S * CHANNELS * N + out_values) + 4.0 * 1024 * 128
    point = {"workload": f"2 ch 44100 -> 47999 Hz, {S} streams x {N} frames, 512-value calls", "kernel": "fir_generic_bulk_kernel (reference two-row form, f32)",
             "kernel_variant": handles[0].kernel_variant(), "step_ms_wall": round(wall_ms, 4),
             "frac_wall": round(alg / (wall_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
             "Msamples_in_per_s": round(S * CHANNELS * N / (wall_ms * 1e-3) / 1e6, 1)}
    del batch, handles
    return point


def fir_pcm_point(ctx: Ctx, ra, args, bits: int, steps: int):
    """The headline launch fed 16-bit WAV samples as they are in the file (rsmp_fir_batch_resample_bulk_pcm_device: the
    conversion of resample/src/main.rs:128-137 inside the kernels' loads) against the two-pass route: the conversion
    pass (rsmp_pcm_to_stereo_f32_device per stream) + the f32 launch."""
    torch = ctx.torch
    S, N = args.streams, args.frames
    handles = [ra.ResamplerFir.new(CHANNELS, ra.SampleRate.Hz44100, ra.SampleRate.Hz48000, ra.Latency.Sample64,
                                   ra.Attenuation.Db90, device=ctx.local_rank) for _ in range(S)]
    dt = {16: torch.int16, 32: torch.int32}[bits]
    lim = 1 << (bits - 2)
    pcm = [torch.randint(-lim, lim, (CHANNELS * N,), device=ctx.dev, dtype=dt).view(torch.uint8) for _ in range(S)]
    cap = handles[0].bulk_output_bound(CHANNELS * N, args.chunk)
    d_out = [torch.empty(cap, device=ctx.dev, dtype=torch.float32) for _ in range(S)]
    f32 = [torch.empty(CHANNELS * N, device=ctx.dev, dtype=torch.float32) for _ in range(S)]
    batch = ra.FirBatch(handles)

    def step():
        batch.reset()
        return batch.resample_bulk_pcm_device(pcm, bits, d_out, args.chunk, ctx.stream)
    _, produced = step()
    out_values = int(sum(produced))
    spinup(ctx, step, 0.3)
    handles[0].set_profiling(True)
    for _ in range(steps):
        step()
    k_ms, _ = handles[0].mean_kernel_ms()
    handles[0].set_profiling(False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        e0.record(torch.cuda.current_stream())
        for p, f in zip(pcm, f32):
            ra.pcm_to_stereo_f32_device(p, bits, CHANNELS, f, ctx.stream)
        e1.record(torch.cuda.current_stream())
        e1.synchronize()
    conv_ms = e0.elapsed_time(e1)
    alg = (bits // 8) * S * CHANNELS * N + 4.0 * out_values + 4.0 * 1024 * 128
    ach = alg / (k_ms * 1e-3) / 1e9
    point = {"kernel": "fir_split_kernel (fp16x2 MFMA, PCM%d loads)" % bits, "kernel_ms": round(k_ms, 4),
             "algorithmic_bytes": int(alg), "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4),
             "Msamples_in_per_s_kernel": round(S * CHANNELS * N / (k_ms * 1e-3) / 1e6, 1),
             "two_pass_conversion_ms": round(conv_ms, 4),
             "what": "the headline launch reading %d-bit PCM in place; two_pass_conversion_ms = what the separate conversion pass "
                     "of the same %d streams costs in front of the f32 launch" % (bits, S)}
    del batch, handles
    return point


def bench_fft(ctx: Ctx, args, steps: int, warmup: int, with_cpu: bool):
    """ResamplerFft 2 ch 44.1k -> 48k (BASELINE config 3): a batch of `--streams` streams x 892 blocks
    of 1176 frames per step, one launch of the overlap-add FFT kernel."""
    import resampler_amd as ra
    from resampler_amd import synth
    torch = ctx.torch
    S, blocks = args.streams, 892
    hs = [ra.ResamplerFft.new(CHANNELS, ra.SampleRate.Hz44100, ra.SampleRate.Hz48000, device=ctx.local_rank)
          for _ in range(S)]
    n_in, n_out = hs[0].chunk_size_input(), hs[0].chunk_size_output()
    base = torch.from_numpy(synth.sweep(blocks * n_in // CHANNELS, CHANNELS, float(IN_HZ))).to(ctx.dev)
    gains = torch.linspace(0.5, 1.0, S, device=ctx.dev)
    d_in = [(base * gains[i]).contiguous() for i in range(S)]
    d_out = [torch.empty(blocks * n_out, device=ctx.dev, dtype=torch.float32) for _ in range(S)]
    batch = ra.FftBatch(hs)
    batch.bind(d_in, d_out, [blocks] * S)

    def step():
        batch.resample_bulk_device(ctx.stream)
    step()
    spinup(ctx, step, args.spinup_seconds)
    dt, _ = timed(ctx, step, steps, max(1, warmup))
    hs[0].set_profiling(True)
    k = []
    for _ in range(32):
        step()
        k.append(hs[0].last_kernel_ms())
    hs[0].set_profiling(False)
    k_ms = float(np.mean(k))
    values_in = S * blocks * n_in
    alg_bytes = 4.0 * S * blocks * (n_in + n_out)
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    line = {
        "metric": "Msamples/s (in) 44.1k->48k FFT overlap-add",
        "value": round(values_in * ctx.world * steps / dt / 1e6, 1), "unit": "Msamples/s",
        "n_gpus": ctx.world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(dt / steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"ResamplerFft 2ch 44100->48000, {S} streams/GPU x {blocks} blocks of "
                               f"1176 frames per step, one launch per step"},
        "roofline": {"bound": "hbm", "kernel": "fft_ola_pair_kernel (wave per two-channel stream, 1176/1280 plan)",
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic_from_profiles("fft"),
                     "kernel_ms": round(k_ms, 4), "kernel_ms_median": round(float(np.median(k)), 4),
                     "kernel_ms_max": round(float(np.max(k)), 4), "kernel_launches_timed": len(k),
                     "algorithmic_bytes": int(alg_bytes)},
    }
    copy_gbs = device_copy_gbs(ctx, alg_bytes / 2)
    line["roofline"]["device_copy"] = round(copy_gbs, 1)
    line["roofline"]["frac_of_device_copy"] = round(achieved / copy_gbs, 4)
    if with_cpu:
        line["cpu_baseline"] = cpu_baseline_fft(min(args.cpu_seconds, 6.0), min(args.cpu_all_cores_seconds, 4.0))
    return line


def bench_c4(ctx: Ctx, args, steps: int, warmup: int):
    """BASELINE config 4: `--c4-streams` independent 2-channel ResamplerFir streams, stream i = ordered
    pair i mod 6 of the 44.1k / 48k / 96k conversions, lock-step steps of 512 frames; the batch is
    partitioned over the ranks by predicted work (strong scaling: the batch is fixed, the ranks share
    it).  A step = one 512-frame chunk for every stream = one launch of fir_lockstep_kernel per GPU.
    --feed rccl: the chunks of a step sit on GPU 0 and travel to their GPU by RCCL send/recv
    (scatter-v), the outputs travel back (gather-v); default: chunks are resident on their GPU."""
    import resampler_amd as ra
    from resampler_amd import sharding, synth
    torch = ctx.torch
    n, frames = args.c4_streams, 512
    specs = sharding.mixed_rate_batch(n, CHANNELS, frames)
    lo, hi = sharding.shard(specs, ctx.rank, ctx.world)
    mine = specs[lo:hi]
    hs = [ra.ResamplerFir.new_from_hz(s.channels, s.in_hz, s.out_hz, ra.Latency.Sample64,
                                      ra.Attenuation.Db90, device=ctx.local_rank) for s in mine]
    caps_all = [sharding.buffer_size_output(s) for s in specs]
    caps = caps_all[lo:hi]
    assert all(c == h.buffer_size_output() for c, h in zip(caps, hs))
    ls = ra.FirLockstep(hs, frames) if hs else None
    feed = None
    kk = max(1, args.c4_k)                          # calls per launch (rsmp_fir_lockstep_run)
    if args.feed == "rccl":
        kk = 1                                      # (the exchange moves one step's chunks per group)
    ring = max(8, kk)                               # chunks of input resident per stream, cycled
    if args.feed == "rccl":
        # a step's chunks arrive from GPU 0 and its outputs return there: the streams are bound straight
        # to their slices of the exchange buffers
        parts = sharding.partition([s.work() for s in specs], ctx.world)
        feed = sharding.StepFeed(ctx.dist, ctx.rank, ctx.world, parts, [frames * CHANNELS] * n, caps_all, ctx.dev,
                                 loopback=(ctx.world == 1))
        stage_in = stage_out = None
        if ctx.rank == 0:
            stage_in = [torch.from_numpy(synth.fast_noise(n * frames * CHANNELS, seed=10 + j)).to(ctx.dev) for j in range(ring)]
            stage_out = torch.empty(sum(caps_all), device=ctx.dev, dtype=torch.float32)
        d_in = [feed.local_in_view(i) for i in range(lo, hi)]
        d_out = [feed.local_out_view(i) for i in range(lo, hi)]
        ring_frames = 0
    else:
        x = torch.from_numpy(synth.fast_noise(ring * frames * CHANNELS, seed=1 + ctx.rank)).to(ctx.dev)
        gains = torch.linspace(0.5, 1.0, max(1, len(mine)), device=ctx.dev)
        d_in = [(x * gains[i]).contiguous() for i in range(len(mine))]
        # (a run of kk calls appends their outputs: at most ceil(512 * out / in) + 1 frames per call)
        d_out = [torch.empty(c + (kk - 1) * CHANNELS * (frames * s.out_hz // s.in_hz + 2), device=ctx.dev, dtype=torch.float32)
                 for c, s in zip(caps, mine)]
        ring_frames = frames
    if ls:
        ls.bind_caps(d_in, d_out, caps)
    k = [0]

    def step():
        if feed:
            feed.scatter(stage_in[k[0] % ring] if ctx.rank == 0 else None)
        if ls and kk > 1:    # kk calls per stream in one go, from the front of the resident input
            ls.run(kk, frames, 0, append=False, stream=ctx.stream)
        elif ls:
            ls.step(frames, (k[0] % ring) * ring_frames, append=False, stream=ctx.stream)
        if feed:
            feed.gather(stage_out)
        k[0] += 1
    step()
    spinup(ctx, step, args.spinup_seconds)
    launches, warm_launches = max(1, steps // kk), max(1, warmup // kk) if warmup else 0
    steps = launches * kk
    st0 = ls.stats() if ls else {}
    dt, host_dt = timed(ctx, step, launches, warm_launches)
    st1 = ls.stats() if ls else {}
    k_ms = k_med = k_max = 0.0
    out_values = 0
    if ls:
        # device time of every launch of a second pass (HIP events on the launch stream, no host sync between the
        # launches): mean -> roofline, median and max beside it (one slow launch moves a mean of few)
        n_prof = min(256, max(launches, 2, 32 // kk))
        ls.set_profiling(True)
        for _ in range(n_prof):
            step()
        per_launch = ls.kernel_ms(n_prof) / kk     # per 512-frame step
        ls.set_profiling(False)
        k_ms, k_med, k_max = float(per_launch.mean()), float(np.median(per_launch)), float(per_launch.max())
        if kk > 1:
            _, prod = ls.run_counts()
            out_values = int(prod.sum()) // kk
        else:
            _, prod = ls.counts()
            out_values = int(prod.sum())
    k_ms = ctx.max_over_ranks(k_ms)
    out_values_all = ctx.sum_over_ranks(float(out_values))
    values_in = n * CHANNELS * frames
    # algorithmic bytes of one step: every input value read once, every output value written once
    alg = 4.0 * (values_in + out_values_all)
    ach = alg / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    return {
        "metric": "Msamples/s (in) config 4: 1024 mixed-rate FIR streams, 512-frame lock-step steps"
                  + (f", {kk} steps per launch" if kk > 1 else ", one step per launch"),
        "value": round(values_in * steps / dt / 1e6, 1), "unit": "Msamples/s", "n_gpus": ctx.world,
        "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 5),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": ("f32 (fp16x2-split products, f32 accumulate)" if ls and ls.split_workgroups() == ls.workgroups()
                  else "f32 (fp16x2-split products, f32 accumulate; exact-f32 products for %d of %d workgroups)"
                  % (ls.workgroups() - ls.split_workgroups(), ls.workgroups()) if ls and ls.split_workgroups()
                  else "f32 (exact-f32 MFMA)"), "data": "synthetic",
        "config": {"workload": f"{n} ResamplerFir streams 2ch 128-tap (Sample64/Db90), pairs 44.1/48/96 kHz "
                               f"(6 ordered), {frames}-frame lock-step steps on carried state, "
                               + ("one launch per step and GPU" if kk == 1 else
                                  f"{kk} steps per launch (rsmp_fir_lockstep_run: planned on the device, one bulk launch per rate pair)")
                               + ", streams partitioned by predicted work",
                   "steps_per_launch": kk, "streams_this_rank": len(mine), "workgroups_this_rank": ls.workgroups() if ls else 0,
                   "feed": ("rccl send/recv scatter-v + gather-v through GPU 0, inside the timed step"
                            + (" (world of one rank: GPU 0 sends to and receives from itself)" if ctx.world == 1 else ""))
                           if feed else "resident per GPU (no data-path collective)",
                   "host_enqueue_ms_per_step": round(host_dt / steps * 1e3, 5),
                   "launches_timed": launches, "timed_region_ms": round(dt * 1e3, 2),
                   # what the batch did inside the timed region (rsmp_fir_lockstep_stats): class tables replaced because the
                   # streams' f64 drift had moved on, runs whose plan was made ahead on the plan stream
                   "table_rebinds_timed": st1.get("table_rebinds", 0) - st0.get("table_rebinds", 0),
                   "plan_ahead_hits_timed": st1.get("plan_ahead_hits", 0) - st0.get("plan_ahead_hits", 0),
                   "table_waits_total": st1.get("table_waits", 0), "late_table_polls_total": st1.get("late_table_polls", 0)},
        "roofline": {"bound": "hbm", "kernel": ("fir_lockstep_kernel (%s, row = stream)" %
                               ("fp16x2 MFMA" if ls and ls.split_workgroups() else "exact-f32 MFMA")) if kk == 1 else
                               "fir_lockstep_plan_kernel + fir_split_kernel per rate pair (whole run, per 512-frame step)",
                     "achieved": round(ach, 1), "peak": HBM_PEAK_GBS * ctx.world, "unit": "GB/s",
                     "frac": round(ach / (HBM_PEAK_GBS * ctx.world), 4), "traffic": traffic_from_profiles("c4"),
                     "kernel_ms": round(k_ms, 5), "kernel_ms_median": round(k_med, 5), "kernel_ms_max": round(k_max, 5),
                     "kernel_ms_covers": ("the step kernel" if kk == 1 else
                                          "the caller's stream from the run's first launch to its last, per 512-frame step "
                                          "(the planner of a run planned ahead ran on the plan stream beside the run before)"),
                     "algorithmic_bytes": int(alg)},
    }


def bench_fir(ctx: Ctx, args):
    import resampler_amd as ra
    S, N = args.streams, args.frames
    kernel = {"auto": ra.FirKernel.Auto, "generic": ra.FirKernel.Generic,
              "periodic": ra.FirKernel.Periodic, "periodic-vector": ra.FirKernel.PeriodicVector,
              "periodic-f32": ra.FirKernel.PeriodicF32}[args.kernel]
    handles, batch = make_fir_batch(ctx, ra, S, N, args.chunk, kernel)

    def step():
        batch.reset()                     # a fresh stream per step (a new file)
        return batch.resample_bulk_device(args.chunk, ctx.stream)

    t_plan0 = time.perf_counter()
    consumed, produced = step()          # first step also builds the plan / class table
    ctx.torch.cuda.synchronize()
    plan_cold_ms = (time.perf_counter() - t_plan0) * 1e3
    assert all(c == CHANNELS * N for c in consumed), "bulk call must consume every frame"
    produced = np.array(produced).copy()
    spinup(ctx, step, args.spinup_seconds)
    # HIP events bracket the convolution launch of every timed step on the launch stream (recorded
    # inside the library, no host sync between steps): roofline.achieved uses their mean.
    for _ in range(max(0, args.warmup - 1)):
        step()
    handles[0].set_profiling(True)
    dt, host_dt = timed(ctx, step, args.steps, 1 if args.warmup > 0 else 0)
    k_ms, k_launches = handles[0].mean_kernel_ms()
    handles[0].set_profiling(False)
    # what a step costs the host when it does not wait for the GPU (in the timed loop the enqueue runs at
    # most a launch or two ahead, so `host_enqueue_ms_per_step` there is mostly back-pressure)
    host_idle = 0.0
    for _ in range(10):
        ctx.torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        host_idle += time.perf_counter() - t1
    ctx.torch.cuda.synchronize()

    values_in_per_step = S * CHANNELS * N            # per rank
    values_out_per_step = int(produced.sum())
    alg_bytes = 4.0 * (values_in_per_step + values_out_per_step) + 4.0 * 1024 * 128
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    copy_gbs = device_copy_gbs(ctx, alg_bytes / 2)
    variant = handles[0].kernel_variant()
    line = {
        "metric": "Msamples/s (in) 44.1k->48k FIR 128-tap",
        "value": round(values_in_per_step * ctx.world * args.steps / dt / 1e6, 1),
        "unit": "Msamples/s",
        "n_gpus": ctx.world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": FIR_DTYPE.get(variant, "f32"),
        "data": "synthetic",
        "config": {
            "workload": f"ResamplerFir 2ch interleaved 44100->48000, 128-tap (Sample64/Db90), "
                        f"{S} streams/GPU x {N}-frame sine sweep per step, bulk driver loop "
                        f"with {args.chunk}-value calls, one launch per step",
            "streams_per_gpu": S,
            "frames_per_stream": N,
            "kernel": args.kernel,
            "out_values_per_step": values_out_per_step,
            "plan_cold_ms": round(plan_cold_ms, 2),
            "spinup_s": args.spinup_seconds,
            "host_enqueue_ms_per_step": round(host_dt / args.steps * 1e3, 4),
            "host_cost_ms_per_step_gpu_idle": round(host_idle / 10 * 1e3, 4),
            "feed": "resident per GPU (no data-path collective)",
        },
        "roofline": {
            "bound": "hbm",
            "kernel": KERNEL_NAMES.get(variant, "?"),
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic_from_profiles("fir", KERNEL_NAMES.get(variant)),
            "kernel_ms": round(k_ms, 4),
            "kernel_launches_timed": int(k_launches),
            "algorithmic_bytes": int(alg_bytes),
            # a plain device copy of the same volume on this GPU (read + written GB/s) and the kernel against it
            "device_copy": round(copy_gbs, 1),
            "frac_of_device_copy": round(achieved / copy_gbs, 4) if copy_gbs else None,
            # useful f32 FMAs (128 taps per output value), T/s
            "fma_per_s": round(values_out_per_step * 128 / (k_ms * 1e-3) / 1e12, 2),
        },
    }
    del batch, handles
    return line


def secondary_lines(ctx: Ctx, args):
    """N = 1 only: the other figures the headline line is read against."""
    import resampler_amd as ra
    sec = {}

    def guard(name, fn):
        """One secondary figure; a failure costs that figure, not the line (the headline does not depend on any of them)."""
        try:
            sec[name] = fn()
        except Exception as e:
            sec[name] = {"error": repr(e)[:300]}
            ctx.torch.cuda.synchronize()

    def fft_point():
        fft = bench_fft(ctx, args, steps=120, warmup=2, with_cpu=not args.no_cpu)   # (120 launches: ~60 ms)
        return {"metric": fft["metric"], "value": fft["value"], "unit": fft["unit"],
                "ms_per_step": fft["ms_per_step"], "workload": fft["config"]["workload"],
                "roofline": fft["roofline"], "cpu_baseline": fft.get("cpu_baseline")}
    guard("fft", fft_point)
    # (64 profiled launches each: 25-30 ms of kernel time behind half a second of spin-up)
    guard("fir_exact_f32", lambda: fir_kernel_point(ctx, ra, args, ra.FirKernel.PeriodicF32, 64))
    guard("fir_vector_no_mfma", lambda: fir_kernel_point(ctx, ra, args, ra.FirKernel.PeriodicVector, 64))
    # the split kernel with every f32 operand cut EXACTLY into three bf16 planes (six products per term): a knob the
    # library reads once per process, so a child runs the same headline launch with it
    try:
        child = subprocess.run([sys.executable, os.path.abspath(__file__), "--no-cpu", "--no-secondary", "--steps", "16",
                                "--warmup", "3", "--spinup-seconds", "1", "--streams", str(args.streams),
                                "--frames", str(args.frames)],
                               env=dict(os.environ, RSMP_DEBUG="1", RSMP_FIR_SPLIT_PLANES="3", WORLD_SIZE="1", RANK="0", LOCAL_RANK=str(ctx.local_rank)),
                               capture_output=True, text=True, timeout=300)
        r3 = json.loads(child.stdout.strip().splitlines()[-1])["roofline"]
        sec["fir_split_bf16x3"] = {k: r3[k] for k in ("kernel", "kernel_ms", "achieved", "frac")}
    except Exception as e:   # (the headline does not depend on it)
        sec["fir_split_bf16x3"] = {"error": repr(e)[:200]}
    # other channel counts on the default kernel (same rate pair and taps)
    guard("fir_channels", lambda: {str(c): fir_channels_point(ctx, ra, args, c, 48) for c in (1, 3, 4, 6, 8)})
    guard("fir_nonperiodic", lambda: fir_nonperiodic_point(ctx, ra, args))
    guard("fir_pcm16", lambda: fir_pcm_point(ctx, ra, args, 16, 32))
    def distinct_point():
        # the same launch with every stream in a different state: nothing shares a plan
        t0 = time.perf_counter()
        handles, batch = make_fir_batch(ctx, ra, args.streams, args.frames, args.chunk, ra.FirKernel.Auto,
                                        distinct_states=True)
        t1 = time.perf_counter()
        setup_s = t1 - t0
        # (1) the C entry as it is, rsmp_fir_batch_resample_bulk_device: every distinct state planned on the host's worker pool
        batch.device_planner = False
        batch.resample_bulk_device(args.chunk, ctx.stream)
        ctx.torch.cuda.synchronize()
        cold = (time.perf_counter() - t1) * 1e3
        # ... and again and again WITHOUT reset: the streams go on from where they are, every launch plans every stream anew
        # (no plan-cache hit), which is what a service that keeps feeding the same streams sees
        again = []
        handles[0].set_profiling(True)
        for _ in range(6):
            t2 = time.perf_counter()
            batch.resample_bulk_device(args.chunk, ctx.stream)
            ctx.torch.cuda.synchronize()
            again.append((time.perf_counter() - t2) * 1e3)
        k_again, _ = handles[0].mean_kernel_ms()
        handles[0].set_profiling(False)
        host_planned = {"step_ms_cold": round(cold, 2), "step_ms_replanned_median": round(sorted(again)[len(again) // 2], 2),
                        "kernel_ms_replanned": round(k_again, 3),
                        "what": "rsmp_fir_batch_resample_bulk_device_ex with planner = 0 (FirBatch.device_planner = False): every distinct state planned "
                                "on the host's worker pool, counts returned when the launch is enqueued"}
        # (2) the entry as a caller gets it, rsmp_fir_batch_resample_bulk_device: a batch in this many different states goes through the
        # device planner behind the same entry (a lock-step batch over the same handles, kept by the library from launch to launch,
        # the states written back into the handles before the call returns -- so the call returns when the launch is through)
        batch.device_planner = None
        t1 = time.perf_counter()
        batch.resample_bulk_device(args.chunk, ctx.stream)
        ctx.torch.cuda.synchronize()
        cold = (time.perf_counter() - t1) * 1e3
        again = []
        for _ in range(12):   # (the first four find the plan stream and have no run to repeat yet: `step_ms_replanned_each`)
            t2 = time.perf_counter()
            batch.resample_bulk_device(args.chunk, ctx.stream)
            ctx.torch.cuda.synchronize()
            again.append((time.perf_counter() - t2) * 1e3)
        routed = bool(batch.planned_on_device)
        k_again = host_planned["kernel_ms_replanned"]
        # ... and the same batch through the DEVICE planner (rsmp_fir_lockstep_run_bulk: the calls' structure, the f64 chain and
        # the wrapped outputs planned by three small kernels, nothing replayed on the host, no host threads): the first
        # launch, then launches that continue the streams (each planned anew, the next one planned ahead beside this one's
        # bulk kernel).  A stream is 4096 calls here and the chain is serial per stream: the planner, not the kernel, is the
        # launch's length.
        dp = {}
        try:
            frames_call = args.chunk // CHANNELS
            ls = ra.FirLockstep(handles, frames_call)
            caps = [h.buffer_size_output() for h in handles]
            ls.bind_caps(batch._keep[0], batch._keep[1], caps)
            ctx.torch.cuda.synchronize()
            t3 = time.perf_counter()
            ls.run_bulk(args.frames, frames_call, 0, append=False, stream=ctx.stream)
            ctx.torch.cuda.synchronize()
            dp_cold = (time.perf_counter() - t3) * 1e3
            dp_again = []
            for _ in range(8):
                t4 = time.perf_counter()
                ls.run_bulk(args.frames, frames_call, 0, append=False, stream=ctx.stream)
                ctx.torch.cuda.synchronize()
                dp_again.append((time.perf_counter() - t4) * 1e3)
            t5 = time.perf_counter()
            for _ in range(8):   # ... and without a wait in between: the planner of launch r + 1 beside the kernel of launch r
                ls.run_bulk(args.frames, frames_call, 0, append=False, stream=ctx.stream)
            ctx.torch.cuda.synchronize()
            dp = {"step_ms_cold": round(dp_cold, 2), "step_ms_replanned_median": round(sorted(dp_again)[len(dp_again) // 2], 2),
                  "step_ms_back_to_back": round((time.perf_counter() - t5) * 1e3 / 8, 2), "calls_per_stream": args.frames // frames_call,
                  "what": "rsmp_fir_lockstep_run_bulk on the same 64 streams: planned on the device (no host planning, no host threads)"}
            ls.sync()
            ls.close()
        except Exception as e:
            dp = {"error": repr(e)[:200]}
        return {
            "device_planned": dp,
            "host_planned_c_entry": host_planned,
            "routed_through_device_planner": routed,
            "what": f"{args.streams} streams in {args.streams} different states through rsmp_fir_batch_resample_bulk_device: the first launch, then "
                    f"launches that continue the streams (planned anew every time), wall clock of a launch incl. synchronisation.  The entry "
                    f"routes a batch in >= {ra.FirBatch.kDevicePlanStates} different states through rsmp_fir_lockstep_run_bulk over the same handles "
                    f"(planned on the device; `host_planned_c_entry`: the same entry with its host planner named, planner = 0)",
            "step_ms_cold": round(cold, 2), "step_ms_replanned_median": round(sorted(again)[len(again) // 2], 2),
            "step_ms_replanned_each": [round(t, 2) for t in again],
            "kernel_ms_replanned": round(k_again, 3), "setup_s": round(setup_s, 2)}
    guard("fir_distinct_states", distinct_point)
    keys = ("metric", "value", "unit", "ms_per_step", "scaling", "dtype", "config", "roofline")
    # (64 launches of the configuration's 256 steps: ~60 ms; 72 launches of config 5: ~53 ms)
    guard("config4", lambda: {k: v for k, v in bench_c4(ctx, args, steps=256 * 64, warmup=256 * 2).items() if k in keys})
    guard("config5", lambda: {k: v for k, v in bench_c5(ctx, args, steps=72, warmup=2).items() if k in keys})

    def shard_point():
        """Config 4's strong scaling, predicted on one GPU: the shard an 8-GPU run hands a rank (1024 / 8 = 128 streams) takes
        t(128) per step, the whole batch t(1024): 8 GPUs are t(1024) / t(128) times faster than one, at best (no
        data-path collective; VERDICT r04 item 2 asks for the figure in the line).  What keeps it below 8 is the run
        planner's serial chain: a shard of an eighth of the streams has the same 256 calls per stream to walk."""
        import copy
        a2 = copy.copy(args)
        a2.c4_streams = max(1, args.c4_streams // 8)
        shard = bench_c4(ctx, a2, steps=256 * 64, warmup=256 * 2)
        whole = sec.get("config4", {}).get("ms_per_step")
        out = {"streams": a2.c4_streams, "ms_per_step": shard["ms_per_step"], "steps_per_launch": shard["config"]["steps_per_launch"]}
        if whole:
            out["predicted_8gpu_speedup_of_config4"] = round(whole / shard["ms_per_step"], 2)
        return out
    guard("config4_shard_of_8", shard_point)
    return sec


def device_hash_noise(torch, dev, start: int, n_values: int, seed: int = 0):
    """synth.hash_noise (splitmix64 of the value's index in the stream) evaluated on the GPU for values
    [start, start + n_values): every rank can produce its own span of one long stream."""
    out = torch.empty(n_values, device=dev, dtype=torch.float32)
    def i64(v):   # a 64-bit pattern as the int64 torch computes with (wrap-around arithmetic, as numpy's uint64)
        v &= (1 << 64) - 1
        return v - (1 << 64) if v >> 63 else v
    c1, c2, c3 = i64((seed + 1) * 0x9E3779B97F4A7C15), i64(0xBF58476D1CE4E5B9), i64(0x94D049BB133111EB)
    step = 1 << 26
    for a in range(0, n_values, step):
        b = min(n_values, a + step)
        z = torch.arange(start + a, start + b, device=dev, dtype=torch.int64) + c1
        z = (z ^ ((z >> 30) & ((1 << 34) - 1))) * c2
        z = (z ^ ((z >> 27) & ((1 << 37) - 1))) * c3
        z = z ^ ((z >> 31) & ((1 << 33) - 1))
        top = (z >> 40) & ((1 << 24) - 1)
        out[a:b] = top.to(torch.float32) / float(1 << 23) - 1.0
    return out


def bench_c5(ctx: Ctx, args, steps: int, warmup: int):
    """BASELINE config 5: ONE 8-channel 96 kHz -> 44.1 kHz stream (Sample64 / Db120), `--c5-frames` input frames
    (default 57.6 M = 10 minutes), 512-frame calls.  The stream is cut into one run of calls per rank
    (sharding.fir_time_shards: the host mirror's exact state at every cut, the buffered frames as halo); a step =
    every rank seeks its resampler to its cut and resamples its run in one bulk launch (strong scaling, no
    data-path collective: a rank holds its span of the input and keeps its span of the output)."""
    import resampler_amd as ra
    from resampler_amd import sharding
    torch = ctx.torch
    ch, in_hz, out_hz, chunk = 8, 96000, 44100, 512
    frames = args.c5_frames
    t0 = time.perf_counter()
    shards = sharding.fir_time_shards(in_hz, out_hz, ra.Latency.Sample64, frames, chunk, ctx.world)
    plan_s = time.perf_counter() - t0
    s = shards[ctx.rank]
    h = ra.ResamplerFir.new_from_hz(ch, in_hz, out_hz, ra.Latency.Sample64, ra.Attenuation.Db120, device=ctx.local_rank)
    first = s.in_offset - s.history_frames
    d_span = device_hash_noise(torch, ctx.dev, first * ch, (s.history_frames + s.in_frames) * ch, seed=5)
    d_hist, d_in = d_span[:s.history_frames * ch], d_span[s.history_frames * ch:]
    d_out = torch.empty(max(1, s.out_frames) * ch + 64, device=ctx.dev, dtype=torch.float32)

    def step():
        h.seek(s.plan, d_hist, ctx.stream)
        if s.in_frames:
            c, p = h.resample_bulk_device(d_in, d_out, chunk * ch, ctx.stream)
            assert (c, p) == (s.in_frames * ch, s.out_frames * ch)
    step()
    spinup(ctx, step, args.spinup_seconds)
    dt, host_dt = timed(ctx, step, steps, warmup)
    h.set_profiling(True)
    k_list = []
    for _ in range(32):
        step()
        k_list.append(h.last_kernel_ms())
    h.set_profiling(False)
    k_ms = ctx.max_over_ranks(float(np.mean(k_list)))
    # the same stream call by call (what a live 8-channel feed does): latency of one 512-frame resample() on
    # HBM-resident buffers, call + wait, rank 0 only
    lat = None
    if ctx.rank == 0:
        hl = ra.ResamplerFir.new_from_hz(ch, in_hz, out_hz, ra.Latency.Sample64, ra.Attenuation.Db120, device=ctx.local_rank)
        d_chunk = d_span[:chunk * ch].contiguous() if d_span.numel() >= chunk * ch else torch.zeros(chunk * ch, device=ctx.dev)
        d_y = torch.empty(hl.buffer_size_output(), device=ctx.dev, dtype=torch.float32)
        ts = []
        for i in range(520):
            t1 = time.perf_counter()
            hl.resample_device(d_chunk, d_y, ctx.stream)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t1)
        ts = np.array(ts[20:]) * 1e6
        lat = {"p50": round(float(np.percentile(ts, 50)), 1), "p99": round(float(np.percentile(ts, 99)), 1),
               "chunk_frames": chunk, "realtime_factor_p50": round(chunk / in_hz / (float(np.percentile(ts, 50)) * 1e-6), 1)}
    values_in = frames * ch
    out_frames = sum(t.out_frames for t in shards)
    alg = 4.0 * ch * (frames + out_frames)
    ach = alg / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    variant = {0: "fir_generic_kernel", 1: "fir_periodic_kernel (vector)", 2: "fir_periodic_db_kernel (vector)",
               3: "fir_periodic_db_kernel (exact-f32 MFMA)", 4: "fir_split_kernel (bf16x3 MFMA)",
               5: "fir_split_kernel (fp16x2 MFMA)"}.get(h.kernel_variant(), "?")
    return {
        "metric": "Msamples/s (in) config 5: one 8-channel 96k->44.1k FIR stream, time-sharded",
        "value": round(values_in * steps / dt / 1e6, 1), "unit": "Msamples/s", "n_gpus": ctx.world,
        "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": FIR_DTYPE.get(h.kernel_variant(), "f32"),
        "data": "synthetic",
        "config": {"workload": f"ResamplerFir 8ch 96000->44100 128-tap (Sample64/Db120), {frames} input frames in "
                               f"{chunk}-frame calls, cut into {ctx.world} run(s) of calls at the host mirror's exact "
                               f"state, each run one seek + one bulk launch on its GPU",
                   "calls_this_rank": s.n_calls, "halo_frames_this_rank": s.history_frames,
                   "chunk_latency_us": lat, "algorithmic_GBps": round(alg * steps / dt / 1e9, 1),
                   "shard_planning_s": round(plan_s, 3), "host_enqueue_ms_per_step": round(host_dt / steps * 1e3, 4)},
        "roofline": {"bound": "hbm", "kernel": variant, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS * ctx.world,
                     "unit": "GB/s", "frac": round(ach / (HBM_PEAK_GBS * ctx.world), 4),
                     "traffic": traffic_from_profiles("c5", variant) if frames == 57_600_000 and ctx.world == 1 else None,
                     "kernel_variant": h.kernel_variant(),
                     "kernel_ms": round(k_ms, 4), "kernel_ms_median": round(float(np.median(k_list)), 4),
                     "kernel_ms_max": round(float(np.max(k_list)), 4), "kernel_launches_timed": len(k_list),
                     "algorithmic_bytes": int(alg)},
    }


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--path", choices=["fir", "fft"], default="fir",
                    help="fir = headline metric (default); fft = the ResamplerFft line alone")
    ap.add_argument("--config", choices=["c2", "c4", "c5"], default="c2",
                    help="c2 = headline workload (weak scaling); c4 = 1024 mixed-rate lock-step streams (strong); "
                         "c5 = one long 8-channel stream cut into time shards (strong)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=64, help="streams per GPU (c2 / fft)")
    ap.add_argument("--c4-streams", type=int, default=1024, help="streams of the whole config-4 batch")
    ap.add_argument("--c4-k", type=int, default=256, help="config 4: lock-step calls per launch (rsmp_fir_lockstep_run; 256 = the configuration's 256 steps in one go); 1 = one call per launch (rsmp_fir_lockstep_step)")
    ap.add_argument("--frames", type=int, default=1 << 20, help="input frames per stream per step")
    ap.add_argument("--c5-frames", type=int, default=57_600_000, help="input frames of the config-5 stream (10 min at 96 kHz)")
    ap.add_argument("--chunk", type=int, default=512, help="reference call size in f32 values")
    ap.add_argument("--feed", choices=["resident", "rccl"], default="resident",
                    help="c4: rccl = a step's chunks are scattered from GPU 0 and outputs gathered back (RCCL send/recv)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-all-cores-seconds", type=float, default=6.0)
    ap.add_argument("--spinup-seconds", type=float, default=1.0,
                    help="untimed steps run before the --warmup steps until this much time has passed: "
                         "the clock governor needs ~50 ms of load to leave its idle state (0 = off)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--kernel", choices=["auto", "generic", "periodic", "periodic-vector", "periodic-f32"], default="auto")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args)       # nothing below has run: this process never touches the GPU

    # RCCL prints a version banner to the C-level stdout (flushed at exit when stdout is a pipe or a file): everything
    # any library writes to fd 1 goes to stderr, the JSON line alone to the real stdout.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    ctx = Ctx(args)
    if args.path == "fft":
        line = bench_fft(ctx, args, args.steps, args.warmup, with_cpu=(not args.no_cpu and ctx.world == 1))
    elif args.config == "c4":
        line = bench_c4(ctx, args, args.steps, args.warmup)
    elif args.config == "c5":
        line = bench_c5(ctx, args, args.steps, args.warmup)
    else:
        line = bench_fir(ctx, args)
        if ctx.world == 1 and ctx.rank == 0:
            if not args.no_cpu:
                line["cpu_baseline"] = cpu_baseline_fir(args.frames, args.cpu_seconds, args.cpu_all_cores_seconds)
            if not args.no_secondary:
                line["secondary"] = secondary_lines(ctx, args)
                # both arithmetic classes side by side in the headline's own roofline: `frac` is the shipping kernel (f32
                # operands cut into two fp16 planes, f32 accumulation on the matrix cores); the reference's arithmetic
                # class -- every product an f32 FMA -- reaches frac_exact_f32 (matrix cores) / frac_vector_no_mfma
                # (north_star's "no MFMA" form) on the same workload in the same run
                line["roofline"]["frac_exact_f32"] = line["secondary"]["fir_exact_f32"].get("frac")
                line["roofline"]["frac_vector_no_mfma"] = line["secondary"]["fir_vector_no_mfma"].get("frac")
                # the headline's `dtype` is 22-bit block-floating operands; the same launch with every f32
                # operand represented EXACTLY (three bf16 planes, six products per term, f32 accumulation) is carried beside it
                # as a headline of its own, with its own dtype and roofline -- not as a footnote of the first
                b3 = line["secondary"].get("fir_split_bf16x3", {})
                if "frac" in b3:
                    line["headline_exact_operands"] = {
                        "metric": line["metric"], "unit": line["unit"],
                        "value": round(line["value"] * line["roofline"]["kernel_ms"] / b3["kernel_ms"], 1) if b3.get("kernel_ms") else None,
                        "value_basis": "kernel time of the exact-operand build x the default build's step / kernel ratio (same launch, a child process)",
                        "dtype": "f32 (every operand split EXACTLY into three bf16 planes, six partial products per term, f32 accumulate)",
                        "roofline": {"bound": "hbm", "kernel": b3.get("kernel"), "achieved": b3.get("achieved"), "peak": line["roofline"]["peak"],
                                     "unit": "GB/s", "frac": b3.get("frac"), "kernel_ms": b3.get("kernel_ms")},
                        "plain_f32": {"frac_matrix_f32": line["roofline"]["frac_exact_f32"], "frac_vector_no_mfma": line["roofline"]["frac_vector_no_mfma"],
                                      "what": "every product an f32 FMA, as src/fir/avx.rs:25-45: on the f32 matrix instruction / on the vector ALUs alone (north_star's 'no MFMA' form)"},
                    }
    if ctx.rank == 0:
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
