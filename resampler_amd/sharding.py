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
