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


@pytest.mark.parametrize("world,n_streams", [(2, 14), (4, 3)])   # (4 ranks, 3 streams: one rank owns nothing and takes part in no exchange)
def test_world2_gloo_scatter_compute_gather_moves_real_samples(world, n_streams):
    import hashlib

    import torch.multiprocessing as mp
    from oracle import pyoracle as o

    steps, frames = 4, 256
    parts = sharding.partition([s.work() for s in sharding.mixed_rate_batch(n_streams, 2, frames)], world)
    assert (world, n_streams) != (4, 3) or any(a == b for a, b in parts)   # the empty shard this case is about
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


# ---- one long stream cut over several ranks (time shards with a halo, FFT block shards) ---------

def _oracle_loop(ref, x, ch, chunk_frames, max_calls=None):
    """The CLI driver loop (resample/src/main.rs:226-254) on the oracle: returns the outputs and the call count."""
    out = np.zeros(ref.buffer_size_output(), np.float32)
    ys, off, calls = [], 0, 0
    while off < x.size and (max_calls is None or calls < max_calls):
        rc, c, p = ref.resample(x[off:off + chunk_frames * ch], out)
        assert rc == 0
        ys.append(out[:p].copy())
        off += c
        calls += 1
        if c == 0:
            break
    return (np.concatenate(ys) if ys else np.zeros(0, np.float32)), off, calls


@pytest.mark.parametrize("ch,in_hz,out_hz,taps,frames,chunk,world", [
    (2, 44100, 48000, 128, 50_000, 512, 4),
    (8, 96000, 44100, 128, 40_000, 512, 3),      # BASELINE config 5's stream, shortened
    (1, 48000, 44100, 64, 30_011, 500, 5),       # ragged last call
    (2, 44100, 96000, 32, 9_000, 333, 2),
])
def test_time_shards_stand_where_the_reference_stands(ch, in_hz, out_hz, taps, frames, chunk, world):
    """Every cut of sharding.fir_time_shards is in the state the oracle is in after the calls before it, the
    offsets are the oracle's running totals, and a piece started there with ONLY its halo produces the
    oracle's outputs for that piece, bit for bit."""
    import resampler_amd as ra
    from oracle import pyoracle as o
    from resampler_amd import synth

    lat = {16: ra.Latency.Sample8, 32: ra.Latency.Sample16, 64: ra.Latency.Sample32, 128: ra.Latency.Sample64}[taps]
    x = synth.sweep(frames, ch, float(in_hz))
    shards = sharding.fir_time_shards(in_hz, out_hz, lat, frames, chunk, world)
    assert [s.rank for s in shards] == list(range(world))
    assert shards[0].in_offset == 0 and shards[0].out_offset == 0 and shards[0].history_frames == 0
    assert sum(s.in_frames for s in shards) == frames
    whole, consumed, calls = _oracle_loop(o.OracleFir(ch, in_hz, out_hz, taps, 90), x, ch, chunk)
    assert consumed == x.size and calls == sum(s.n_calls for s in shards)
    assert sum(s.out_frames for s in shards) * ch == whole.size
    ref = o.OracleFir(ch, in_hz, out_hz, taps, 90)
    off = 0
    for s in shards:
        assert s.plan.state() == ref.state(), s.rank                     # read_position, available_frames, position
        assert s.in_offset * ch == off and s.history_frames == ref.state()[1]
        piece = o.OracleFir(ch, in_hz, out_hz, taps, 90)
        piece.seek(s.plan.state(), x[(s.in_offset - s.history_frames) * ch:s.in_offset * ch])
        y, c, k = _oracle_loop(piece, x[s.in_offset * ch:(s.in_offset + s.in_frames) * ch], ch, chunk)
        assert k == s.n_calls and c == s.in_frames * ch and y.size == s.out_frames * ch
        assert np.array_equal(y, whole[s.out_offset * ch:(s.out_offset + s.out_frames) * ch]), s.rank
        _, c2, _ = _oracle_loop(ref, x[off:], ch, chunk, s.n_calls)       # advance the sequential run to the next cut
        off += c2


def _time_shard_worker(rank, world, port, ch, in_hz, out_hz, frames, chunk, q):
    """A rank resamples its time shard of ONE stream and the pieces are gathered on rank 0 (gloo; on the GPU
    box the piece runs through ResamplerFir.seek + resample_bulk_device, tests/test_fir_gpu.py)."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import resampler_amd as ra
    from oracle import pyoracle as o
    from resampler_amd import sharding as sh
    from resampler_amd import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = synth.sweep(frames, ch, float(in_hz))                # every rank holds (here: regenerates) its part of the input
    shards = sh.fir_time_shards(in_hz, out_hz, ra.Latency.Sample64, frames, chunk, world)
    s = shards[rank]
    piece = o.OracleFir(ch, in_hz, out_hz, 128, 90)
    piece.seek(s.plan.state(), x[(s.in_offset - s.history_frames) * ch:s.in_offset * ch])
    y, _, _ = _oracle_loop(piece, x[s.in_offset * ch:(s.in_offset + s.in_frames) * ch], ch, chunk)
    assert y.size == s.out_frames * ch
    total = sum(t.out_frames for t in shards) * ch
    if rank == 0:
        out = torch.zeros(total)
        out[:y.size] = torch.from_numpy(y)
        for t in shards[1:]:
            if t.out_frames:
                dist.recv(out[t.out_offset * ch:(t.out_offset + t.out_frames) * ch], src=t.rank)
        q.put(out.numpy())
    elif y.size:
        dist.send(torch.from_numpy(y), dst=0)
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo_time_sharded_stream_equals_the_single_pass():
    import torch.multiprocessing as mp
    from oracle import pyoracle as o
    from resampler_amd import synth

    world, ch, in_hz, out_hz, frames, chunk = 2, 8, 96000, 44100, 30_000, 512
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_time_shard_worker, args=(r, world, port, ch, in_hz, out_hz, frames, chunk, q))
             for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole, _, _ = _oracle_loop(o.OracleFir(ch, in_hz, out_hz, 128, 90), synth.sweep(frames, ch, float(in_hz)), ch, chunk)
    assert np.array_equal(got, whole)


def test_fft_block_shards_need_one_block_of_halo():
    """Blocks [first, end) of a stream on a fresh resampler, preceded by block first - 1 whose output is dropped,
    equal the single pass bit for bit (the overlap reaches back exactly one block, resampler_fft.rs:416-423)."""
    from oracle import pyoracle as o
    from resampler_amd import synth

    ch, in_hz, out_hz, blocks, world = 2, 44100, 48000, 23, 4
    ref = o.OracleFft(ch, in_hz, out_hz)
    n_in, n_out = ref.chunk_size_input(), ref.chunk_size_output()
    x = synth.sweep(blocks * n_in // ch, ch, float(in_hz))
    whole = np.zeros((blocks, n_out), np.float32)
    for b in range(blocks):
        assert ref.resample(x[b * n_in:(b + 1) * n_in], whole[b]) == 0
    ranges = sharding.fft_block_shards(blocks, world)
    assert ranges[0][0] == 0 and ranges[-1][1] == blocks and all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
    for first, end in ranges:
        piece = o.OracleFft(ch, in_hz, out_hz)
        out = np.zeros(n_out, np.float32)
        for b in range(max(first - 1, 0), end):
            assert piece.resample(x[b * n_in:(b + 1) * n_in], out) == 0
            if b >= first:
                assert np.array_equal(out, whole[b]), (first, b)
