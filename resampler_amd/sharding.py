"""Sharding of independent resampler streams across the GPUs of a node (one process per GPU).

The hot path has no exchange step: streams (and channels within a stream) never talk to each other
(reference: one `&mut self` instance per stream, src/resampler_fir.rs:509-513).  So a batch is
split into contiguous ranges of streams, one range per rank, balanced by predicted work; every rank
runs its range through `FirBatch` / `FftBatch` on its own device and only *metadata* (counts) is
ever gathered.  No data-path collective exists, by design; RCCL is used only where the caller
wants results or inputs moved between GPUs ("Multi-GPU").
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence, Tuple


@dataclass(frozen=True)
class StreamSpec:
    """One ResamplerFir stream of a batch."""
    channels: int
    in_hz: int
    out_hz: int
    taps: int = 128
    frames: int = 512          # input frames per step

    def work(self) -> float:
        """Predicted cost of one step: output frames x taps x channels (FMAs of the periodic kernel)."""
        return self.frames * (self.out_hz / self.in_hz) * self.taps * self.channels


def buffer_size_output(spec: "StreamSpec") -> int:
    """ResamplerFir::buffer_size_output (resampler_fir.rs:456-465) in f32 values, without a handle."""
    import math
    return (int(math.ceil((4096 - spec.taps) / (spec.in_hz / spec.out_hz))) + 2) * spec.channels


def partition(weights: Sequence[float], world: int) -> List[Tuple[int, int]]:
    """Contiguous ranges [start, end) per rank with near-equal cumulative weight.

    Deterministic, identical on every rank (pure function of its arguments): rank r takes the
    streams whose cumulative-weight midpoint falls into the r-th of `world` equal slices.
    """
    n = len(weights)
    total = float(sum(weights))
    if world <= 0:
        raise ValueError("world must be positive")
    if n == 0 or total <= 0.0:
        base = [(min(r * ((n + world - 1) // max(world, 1)), n),
                 min((r + 1) * ((n + world - 1) // max(world, 1)), n)) for r in range(world)]
        return base
    bounds = [0]
    acc = 0.0
    rank = 1
    for i, w in enumerate(weights):
        mid = acc + 0.5 * w
        while rank < world and mid >= total * rank / world:
            bounds.append(i)
            rank += 1
        acc += w
    while len(bounds) < world:
        bounds.append(n)
    bounds.append(n)
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


def shard(specs: Sequence[StreamSpec], rank: int, world: int) -> Tuple[int, int]:
    """The contiguous range of `specs` that `rank` owns."""
    return partition([s.work() for s in specs], world)[rank]


def mixed_rate_batch(n_streams: int, channels: int = 2, frames: int = 512) -> List[StreamSpec]:
    """BASELINE config 4: stream i uses ordered pair i mod 6 of the 44.1k / 48k / 96k conversions."""
    pairs = [(44100, 48000), (48000, 44100), (44100, 96000), (96000, 44100), (48000, 96000), (96000, 48000)]
    return [StreamSpec(channels, *pairs[i % 6], 128, frames) for i in range(n_streams)]


class StepFeed:
    """Scatter-v of a step's input chunks from a staging rank to the ranks that own the streams, and
    gather-v of the outputs back: the only exchange the path has.  Built from one group of
    point-to-point sends / receives per direction (`torch.distributed.batch_isend_irecv`: on the `nccl`
    backend = RCCL that is ncclGroupStart + ncclSend / ncclRecv over xGMI); backend agnostic, so the
    world-2 `gloo` test on CPU drives exactly this code.

    in_sizes[i] / out_sizes[i]: f32 values of stream i's input chunk / output room per step; `parts`:
    the contiguous stream range of every rank (sharding.partition)."""

    def __init__(self, dist, rank: int, world: int, parts, in_sizes, out_sizes, device, root: int = 0,
                 loopback: bool = False):
        """loopback: the root's own piece also travels through the process group (a send to itself and the
        matching receive in the same group) instead of a device copy -- lets a single-GPU box execute the
        RCCL send / recv path (bench.py --config c4 --feed rccl --gpus 1)."""
        import torch
        self.dist, self.rank, self.world, self.root = dist, rank, world, root
        self.loopback = bool(loopback and dist is not None)
        self.parts = list(parts)
        self.in_off = [0]
        self.out_off = [0]
        for a, b in zip(in_sizes, out_sizes):
            self.in_off.append(self.in_off[-1] + int(a))
            self.out_off.append(self.out_off[-1] + int(b))
        lo, hi = self.parts[rank]
        self.local_in = torch.empty(self.in_off[hi] - self.in_off[lo], dtype=torch.float32, device=device)
        self.local_out = torch.empty(self.out_off[hi] - self.out_off[lo], dtype=torch.float32, device=device)

    def _slice(self, flat, off, r):
        lo, hi = self.parts[r]
        return flat[off[lo]:off[hi]]

    def _exchange(self, ops):
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()

    def local_in_view(self, i: int):
        """Stream i's chunk inside this rank's flat input buffer (i = global stream index)."""
        lo, _ = self.parts[self.rank]
        return self.local_in[self.in_off[i] - self.in_off[lo]:self.in_off[i + 1] - self.in_off[lo]]

    def local_out_view(self, i: int):
        lo, _ = self.parts[self.rank]
        return self.local_out[self.out_off[i] - self.out_off[lo]:self.out_off[i + 1] - self.out_off[lo]]

    def scatter(self, stage_in=None):
        """root: stage_in = all streams' chunks back to back.  Every rank: fills self.local_in."""
        d = self.dist
        ops = []
        if self.rank == self.root:
            for r in range(self.world):
                piece = self._slice(stage_in, self.in_off, r)
                if r == self.root and not self.loopback:
                    self.local_in.copy_(piece)
                elif piece.numel():
                    ops.append(d.P2POp(d.isend, piece, r))
                    if r == self.root:
                        ops.append(d.P2POp(d.irecv, self.local_in, self.root))
        elif self.local_in.numel():
            ops.append(d.P2POp(d.irecv, self.local_in, self.root))
        self._exchange(ops)
        return self.local_in

    def gather(self, stage_out=None):
        """Every rank: sends self.local_out.  root: stage_out receives all streams' output rooms."""
        d = self.dist
        ops = []
        if self.rank == self.root:
            for r in range(self.world):
                piece = self._slice(stage_out, self.out_off, r)
                if r == self.root and not self.loopback:
                    piece.copy_(self.local_out)
                elif piece.numel():
                    ops.append(d.P2POp(d.irecv, piece, r))
                    if r == self.root:
                        ops.append(d.P2POp(d.isend, self.local_out, self.root))
        elif self.local_out.numel():
            ops.append(d.P2POp(d.isend, self.local_out, self.root))
        self._exchange(ops)
        return stage_out


# ---- one long stream over several GPUs (time shards, block shards) -----------------------------

@dataclass
class TimeShard:
    """A run of consecutive resample() calls of one ResamplerFir stream (frames, not values)."""
    rank: int
    first_call: int
    n_calls: int
    in_offset: int        # the shard's first input frame in the stream
    in_frames: int
    out_offset: int       # its first output frame in the stream's output
    out_frames: int
    history_frames: int   # frames the reference holds buffered when the shard starts: input[in_offset - h : in_offset]
    plan: object          # FirPlan standing at the shard's start (ResamplerFir.seek)


def fir_time_shards(in_hz: int, out_hz: int, latency, in_frames: int, chunk_frames: int, world: int) -> List[TimeShard]:
    """Cuts the CLI driver loop (resample/src/main.rs:226-254) over ``in_frames`` of one stream into ``world``
    runs of calls.  The host mirror (FirPlan: the reference's f64 position recurrence replayed exactly, no
    samples) gives, for every cut, the state the reference is in, how much input it has consumed and how much
    output it has produced; a rank starts its resampler there (``ResamplerFir.seek`` with the ``history_frames``
    input frames before the cut as the halo) and produces exactly the outputs the single pass would.
    Deterministic: every rank computes the same list."""
    from . import FirPlan
    if world <= 0 or chunk_frames <= 0:
        raise ValueError("world and chunk_frames must be positive")
    plan = FirPlan(in_hz, out_hz, latency)
    _, _, total_calls = plan.clone().bulk(in_frames, chunk_frames)
    shards: List[TimeShard] = []
    in_off = out_off = 0
    for r in range(world):
        k0, k1 = (r * total_calls) // world, ((r + 1) * total_calls) // world
        start = plan.clone()
        hist = start.state()[1]
        accepted = produced = calls = 0
        if k1 > k0:
            accepted, produced, calls = plan.bulk(in_frames - in_off, chunk_frames, k1 - k0)
            if r + 1 < world and (calls != k1 - k0 or accepted != calls * chunk_frames):
                raise ValueError("a call of the stream did not accept its whole chunk: cut it with a smaller chunk")
        shards.append(TimeShard(r, k0, calls, in_off, accepted, out_off, produced, hist, start))
        in_off += accepted
        out_off += produced
    return shards


def run_fir_time_shard(handle, shard: TimeShard, d_stream, d_out, channels: int, chunk_frames: int,
                       stream=None) -> Tuple[int, int]:
    """Runs ``shard`` of the stream ``d_stream`` (a CUDA tensor holding at least the shard's input and the
    ``history_frames`` before it, indexed from the start of the stream) on ``handle``'s GPU; the shard's outputs
    go to ``d_out[0 : out_frames * channels]``.  Returns (consumed, produced) in values."""
    c = channels
    handle.seek(shard.plan, d_stream[(shard.in_offset - shard.history_frames) * c:shard.in_offset * c], stream)
    if shard.in_frames == 0:
        return 0, 0
    return handle.resample_bulk_device(d_stream[shard.in_offset * c:(shard.in_offset + shard.in_frames) * c], d_out,
                                       chunk_frames * c, stream)


def fft_block_shards(n_blocks: int, world: int) -> List[Tuple[int, int]]:
    """[first_block, end_block) per rank for one ResamplerFft stream.  A block's output needs the second half of
    its predecessor's inverse transform (the overlap, resampler_fft.rs:416-423) and nothing older, so a rank
    whose range does not start the stream recomputes ONE block before it and drops that block's output
    (run_fft_block_shard)."""
    if world <= 0:
        raise ValueError("world must be positive")
    return [((r * n_blocks) // world, ((r + 1) * n_blocks) // world) for r in range(world)]


def run_fft_block_shard(handle, first: int, end: int, d_stream, d_work, stream=None):
    """Blocks [first, end) of the stream ``d_stream`` (indexed from the start of the stream) on a FRESH
    ``handle`` (zero overlap).  ``d_work`` holds ``end - first + 1`` output blocks: the halo block's output
    lands in front and is dropped; returns the view of ``d_work`` with the shard's outputs (no copy, so nothing
    has to be ordered against the launch, which is asynchronous on ``stream`` / the handle's stream)."""
    n_in, n_out = handle.chunk_size_input(), handle.chunk_size_output()
    if end <= first:
        return d_work[:0]
    halo = 1 if first > 0 else 0
    n = end - first + halo
    handle.resample_bulk_device(d_stream[(first - halo) * n_in:end * n_in], d_work[:n * n_out], n, stream)
    return d_work[halo * n_out:n * n_out]
