"""N > 1 path on CPU: world_size-2 `gloo` processes shard a mixed batch of streams, replay their
streams' control flow with the host mirror (no GPU needed), and gather only metadata.  Checks that
the shards partition the batch, that work is balanced, and that the gathered (consumed, produced)
totals equal what the oracle yields for every stream."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from resampler_amd import sharding  # noqa: E402


def test_partition_properties():
    specs = sharding.mixed_rate_batch(1024)
    w = [s.work() for s in specs]
    for world in (1, 2, 4, 8):
        parts = sharding.partition(w, world)
        assert parts[0][0] == 0 and parts[-1][1] == len(specs)
        assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
        loads = [sum(w[a:b]) for a, b in parts]
        assert max(loads) / (sum(w) / world) < 1.02
    assert sharding.partition([], 4) == [(0, 0)] * 4
    assert sharding.partition([1.0], 4)[0] == (0, 1) or sum(b - a for a, b in sharding.partition([1.0], 4)) == 1
    assert sum(b - a for a, b in sharding.partition([5.0, 1.0, 1.0], 2)) == 3


def _worker(rank, world, port, steps, n_streams, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    import resampler_amd as ra
    from resampler_amd import sharding as sh
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    specs = sh.mixed_rate_batch(n_streams)
    a, b = sh.shard(specs, rank, world)
    mine = []
    for i in range(a, b):
        s = specs[i]
        plan = ra.FirPlan(s.in_hz, s.out_hz, ra.Latency.Sample64)
        cap = 10 ** 6
        consumed = produced = 0
        for _ in range(steps):
            acc, prod = plan.call(s.frames, cap)
            consumed += acc
            produced += prod
        mine.append((i, consumed * s.channels, produced * s.channels))
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)     # metadata only: there is no data-path collective
    dist.barrier()
    if rank == 0:
        q.put([x for part in gathered for x in part])
    dist.destroy_process_group()


def test_world2_gloo_sharded_counts_match_oracle():
    import torch.multiprocessing as mp
    from oracle import pyoracle as o

    world, steps, n_streams = 2, 6, 24
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, steps, n_streams, q)) for r in range(world)]
    for p in procs:
        p.start()
    result = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(i for i, _, _ in result) == list(range(n_streams))
    specs = sharding.mixed_rate_batch(n_streams)
    for i, consumed, produced in result:
        s = specs[i]
        ref = o.OracleFir(s.channels, s.in_hz, s.out_hz, s.taps, 90)
        out = np.zeros(ref.buffer_size_output(), np.float32)
        c_tot = p_tot = 0
        for _ in range(steps):
            rc, c, p = ref.resample(np.zeros(s.frames * s.channels, np.float32), out)
            assert rc == 0
            c_tot += c
            p_tot += p
        assert (consumed, produced) == (c_tot, p_tot), i


def _feed_worker(rank, world, port, steps, n_streams, frames, q):
    """One rank of the sharded lock-step pipeline on CPU: scatter-v of every step's chunks from rank 0
    (sharding.StepFeed over gloo -- the code bench.py --feed rccl runs over RCCL), one resample() call per
    owned stream and step, gather-v of the outputs back.  No GPU here, so the per-stream call is made by
    the CPU oracle (test infrastructure); the partition, the offsets and the exchange are the product's."""
    sys.path.insert(0, ROOT)
    import hashlib

    import torch
    import torch.distributed as dist

    from oracle import pyoracle as o
    from resampler_amd import sharding as sh
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    specs = sh.mixed_rate_batch(n_streams, 2, frames)
    parts = sh.partition([s.work() for s in specs], world)
    lo, hi = parts[rank]
    refs = {i: o.OracleFir(specs[i].channels, specs[i].in_hz, specs[i].out_hz, specs[i].taps, 90) for i in range(lo, hi)}
    caps = []
    for s in specs:
        r = o.OracleFir(s.channels, s.in_hz, s.out_hz, s.taps, 90)
        caps.append(r.buffer_size_output())
    in_sizes = [frames * s.channels for s in specs]
    feed = sh.StepFeed(dist, rank, world, parts, in_sizes, caps, torch.device("cpu"))
    rng = np.random.default_rng(7)
    x_all = (rng.random((steps, sum(in_sizes)), dtype=np.float32) * 2 - 1).astype(np.float32)   # same on every rank (seeded)
    stage_out = torch.zeros(sum(caps)) if rank == 0 else None
    digests = [hashlib.sha256() for _ in specs] if rank == 0 else None
    totals = [[0, 0] for _ in specs]
    for k in range(steps):
        feed.scatter(torch.from_numpy(x_all[k]) if rank == 0 else None)
        counts = torch.zeros(2 * n_streams, dtype=torch.int64)
        for i in range(lo, hi):
            out = np.zeros(caps[i], np.float32)
            rc, c, p = refs[i].resample(feed.local_in_view(i).numpy(), out)
            assert rc == 0
            feed.local_out_view(i).copy_(torch.from_numpy(out))
            counts[2 * i], counts[2 * i + 1] = c, p
        feed.gather(stage_out)
        dist.all_reduce(counts)          # metadata only
        if rank == 0:
            for i in range(n_streams):
                p = int(counts[2 * i + 1])
                digests[i].update(stage_out[feed.out_off[i]:feed.out_off[i] + p].numpy().tobytes())
                totals[i][0] += int(counts[2 * i])
                totals[i][1] += p
    dist.barrier()
    if rank == 0:
        q.put(([d.hexdigest() for d in digests], totals, x_all))
    dist.destroy_process_group()


def test_world2_gloo_scatter_compute_gather_moves_real_samples():
    import hashlib

    import torch.multiprocessing as mp
    from oracle import pyoracle as o

    world, steps, n_streams, frames = 2, 4, 14, 256
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_feed_worker, args=(r, world, port, steps, n_streams, frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    digests, totals, x_all = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # every stream's gathered samples and counts equal a single-process run of the same calls
    specs = sharding.mixed_rate_batch(n_streams, 2, frames)
    off = 0
    for i, s in enumerate(specs):
        ref = o.OracleFir(s.channels, s.in_hz, s.out_hz, s.taps, 90)
        out = np.zeros(ref.buffer_size_output(), np.float32)
        h = hashlib.sha256()
        c_tot = p_tot = 0
        n = frames * s.channels
        for k in range(steps):
            rc, c, p = ref.resample(x_all[k][off:off + n], out)
            assert rc == 0
            h.update(out[:p].tobytes())
            c_tot += c
            p_tot += p
        off += n
        assert totals[i] == [c_tot, p_tot], i
        assert digests[i] == h.hexdigest(), i


def test_bench_gpus_flag_spawns_or_fails_loudly():
    # `bench.py --gpus N` must never quietly run one rank: without N visible GPUs it refuses (exit code 2, a
    # message on stderr) before anything touches a device.
    import subprocess
    import torch
    n = torch.cuda.device_count() + 2
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--no-cpu"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 2, (p.returncode, p.stderr[-500:])
    assert f"--gpus {n}" in p.stderr and "visible" in p.stderr
    assert p.stdout.strip() == ""
