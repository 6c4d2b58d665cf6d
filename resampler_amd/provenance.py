"""Which kernel sources a committed counter profile belongs to: profiles/traffic_latest.json carries, per workload,
the SHA-256 of the kernel sources its HBM counters were collected with (tools/profile_summary.py); bench.py reports
`roofline.traffic` only while the sources it runs are the same (a kernel change without a fresh
counter pass silently kept the old ratio)."""
from __future__ import annotations

import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
_FIR = ["fir_split.hip", "fir_periodic.hip", "fir_periodic.h", "fir_kernels.h", "fir_nonfinite.h"]
SOURCES = {
    "fir": _FIR,
    "c5": _FIR,
    "c4": _FIR + ["fir_lockstep.hip", "fir_lockstep_run.hip", "fir_lockstep.h", "fir_mirror_core.h", "fir_mirror_fast.h"],
    "fft": ["fft_pair.hip", "fft_wave.hip", "fft_wave_core.h", "fft_kernels.hip", "fft_kernels.h", "fft_butterflies.h", "fft_butterflies_pk.h"],
}


def kernel_sources_sha(workload: str) -> str | None:
    files = SOURCES.get(workload)
    if not files:
        return None
    h = hashlib.sha256()
    for f in files:
        try:
            with open(os.path.join(CSRC, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read() + b"\0")
        except OSError:
            return None
    return h.hexdigest()[:16]
