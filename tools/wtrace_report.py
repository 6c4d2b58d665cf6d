#!/usr/bin/env python3
"""Summarise an RSMP_FIR_WTRACE dump of fir_periodic_kernel (diagnostic; not part of the product).

Each line of the dump is `block wave t:tag ...` with 100 MHz timestamps.  Tags: 1 item begins,
2 first barrier passed / image ready, 3 staging issued, 4 own DMA landed, 5 staging barrier
passed, 6 tile begins, 7 tile done; producers of the double-buffered kernel: 11 wait for a free
image, 12 free, 13 DMA issued, 14 landed.  Prints where the wave time goes.
"""
import sys
from collections import defaultdict

NAMES = {
    (1, 2): "wait: item barrier",
    (2, 3): "issue staging",
    (3, 4): "wait: own DMA",
    (4, 5): "wait: staging barrier",
    (5, 6): "tile setup (first)",
    (6, 7): "taps",
    (7, 8): "epilogue + stores",
    (8, 6): "tile switch",
    (8, 1): "item switch",
    (2, 1): "padding item",
    (5, 1): "item without tile",
    (7, 6): "tile switch",
    (7, 1): "item switch",
    (2, 6): "item setup",
    (6, 20): "unit setup",
    (20, 21): "unit MFMA loop",
    (21, 7): "unit epilogue",
    (2, 7): "consumer: MFMA stream",
    (7, 8): "consumer: signal + stores",
    (8, 1): "consumer: next item",
    (1, 2): "wait: image staged",
    (12, 14): "producer: stage + signal",
    (12, 13): "producer: stage",
    (13, 14): "producer: wrap + signal",
    (14, 11): "producer: next item",
    (11, 12): "producer: wait image free",
    (12, 13): "producer: claim + issue DMA",
    (13, 14): "producer: wait landed",
    (14, 11): "producer: publish",
}


def main(path):
    spans = defaultdict(float)
    counts = defaultdict(int)
    total = 0.0
    waves = 0
    taps = []
    for line in open(path):
        f = line.split()
        ev = [tuple(int(x) for x in e.split(":")) for e in f[2:]]
        if len(ev) < 2:
            continue
        waves += 1
        total += ev[-1][0] - ev[0][0]
        for (t0, a), (t1, b) in zip(ev, ev[1:]):
            spans[(a, b)] += t1 - t0
            counts[(a, b)] += 1
            if (a, b) == (6, 7):
                taps.append(t1 - t0)
    print(f"{waves} waves, mean traced time {total / waves / 100:.1f} us")
    for k, v in sorted(spans.items(), key=lambda kv: -kv[1]):
        print(f"  {NAMES.get(k, str(k)):28s} {100 * v / total:5.1f} %   n={counts[k]:6d}  mean {v / counts[k] / 100:7.2f} us")
    if taps:
        taps.sort()
        n = len(taps)
        print(f"  taps per tile: p10 {taps[n // 10] / 100:.2f}  p50 {taps[n // 2] / 100:.2f}  p90 {taps[9 * n // 10] / 100:.2f} us")


if __name__ == "__main__":
    main(sys.argv[1])
