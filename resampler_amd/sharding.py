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

    def __init__(self, dist, rank: int, world: int, parts, in_sizes, out_sizes, device, root: int = 0):
        import torch
        self.dist, self.rank, self.world, self.root = dist, rank, world, root
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
                if r == self.root:
                    self.local_in.copy_(piece)
                elif piece.numel():
                    ops.append(d.P2POp(d.isend, piece, r))
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
                if r == self.root:
                    piece.copy_(self.local_out)
                elif piece.numel():
                    ops.append(d.P2POp(d.irecv, piece, r))
        elif self.local_out.numel():
            ops.append(d.P2POp(d.isend, self.local_out, self.root))
        self._exchange(ops)
        return stage_out
