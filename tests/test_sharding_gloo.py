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
