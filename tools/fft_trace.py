#!/usr/bin/env python3
"""Wave timeline of fft_ola_wave_kernel on the bench's FFT workload (diagnostic builds of fft_wave.hip:
`make exp EXPFILE=fft_wave.hip EXPS="91 92"`; 91 = start / end of every wave, 92 = also cycles per phase).

  RSMP_AMD_LIB=resampler_amd/libresampler_amd_exp91.so python tools/fft_trace.py [streams] [blocks]

Prints how long the waves live against the launch (residency), how the end times spread -- over the chip, inside a CU,
by the wave's age on its SIMD -- and, for build 92, the share of each phase."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import resampler_amd as ra
from resampler_amd import synth

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 892
dev = torch.device("cuda:0")
hs = [ra.ResamplerFft.new(2, ra.SampleRate.Hz44100, ra.SampleRate.Hz48000) for _ in range(S)]
n_in, n_out = hs[0].chunk_size_input(), hs[0].chunk_size_output()
base = torch.from_numpy(synth.sweep(blocks * n_in // 2, 2, 44100.0)).to(dev)
d_in = [(base * (0.5 + 0.5 * i / S)).contiguous() for i in range(S)]
d_out = [torch.empty(blocks * n_out, device=dev, dtype=torch.float32) for _ in range(S)]
batch = ra.FftBatch(hs)
batch.bind(d_in, d_out, [blocks] * S)
stream = ra.torch_stream()
for _ in range(int(os.environ.get("TRACE_SPINUP", "600"))):
    batch.resample_bulk_device(stream)
torch.cuda.synchronize()
hs[0].set_profiling(True)
batch.resample_bulk_device(stream)
k_ms = hs[0].last_kernel_ms()
hs[0].set_profiling(False)
torch.cuda.synchronize()

L = ra.lib()
fn = L.rsmp_debug_fft_trace
fn.argtypes = [C.c_void_p, C.c_size_t]
fn.restype = C.c_int
buf = np.zeros(4096 * 16, np.uint64)
rc = fn(buf.ctypes.data, buf.size)
assert rc == 0, rc
t = buf.reshape(4096, 16)
t = t[t[:, 1] > 0]
t0, t1 = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
hw, xcc = (t[:, 2] & 0xffffffff).astype(np.int64), (t[:, 2] >> 32).astype(np.int64) & 0xf
nblk = t[:, 3].astype(np.int64)
wave_id, simd, cu, sh, se = hw & 0xf, (hw >> 4) & 3, (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
start, end = t0.min(), t1.max()
span = (end - start) / 100.0   # us (100 MHz)
dur = (t1 - t0) / 100.0
print(f"kernel {k_ms * 1e3:.1f} us by events; waves {len(t)}; first start -> last end {span:.1f} us")
print(f"wave start: p50 {np.percentile(t0 - start, 50) / 100:.1f} us, max {(t0 - start).max() / 100:.1f} us")
print(f"wave life: p5 {np.percentile(dur, 5):.1f}  p50 {np.percentile(dur, 50):.1f}  p95 {np.percentile(dur, 95):.1f}  max {dur.max():.1f} us;  "
      f"mean life / span = {dur.mean() / span:.3f}")
full = nblk == nblk.max()
print(f"blocks per wave: max {nblk.max()}, waves with fewer: {(~full).sum()} (mean {nblk[~full].mean() if (~full).any() else 0:.1f})")
e = (t1 - start) / 100.0
print(f"end of the full-length waves: p5 {np.percentile(e[full], 5):.1f}  p25 {np.percentile(e[full], 25):.1f}  p50 {np.percentile(e[full], 50):.1f}  "
      f"p75 {np.percentile(e[full], 75):.1f}  p95 {np.percentile(e[full], 95):.1f}  max {e[full].max():.1f} us")
print(f"per block of a full-length wave: p50 {np.percentile(dur[full] / nblk[full], 50):.2f} us")
# by XCD
print("end p50 / max by XCD:", "  ".join(f"{x}: {np.percentile(e[(xcc == x) & full], 50):.0f}/{e[(xcc == x) & full].max():.0f}" for x in sorted(set(xcc))))
# inside a CU: spread between its first and last full-length wave
cus = sorted(set(cu_key))
spread = [e[(cu_key == c) & full].max() - e[(cu_key == c) & full].min() for c in cus if ((cu_key == c) & full).sum() > 1]
print(f"CUs {len(cus)}; waves per CU min {min((cu_key == c).sum() for c in cus)} max {max((cu_key == c).sum() for c in cus)}; "
      f"first-to-last end inside a CU: p50 {np.percentile(spread, 50):.1f}  max {max(spread):.1f} us")
last_by_cu = np.array([e[cu_key == c].max() for c in cus])
print(f"last end by CU: p5 {np.percentile(last_by_cu, 5):.1f}  p50 {np.percentile(last_by_cu, 50):.1f}  p95 {np.percentile(last_by_cu, 95):.1f}  max {last_by_cu.max():.1f}")
# by the wave's slot on its SIMD (age order)
for s_ in range(4):
    m = (simd == s_) & full
    ranks = []
    for c in cus:
        mm = m & (cu_key == c)
        if mm.sum() >= 2:
            order = np.argsort(wave_id[mm])
            ranks.append(e[mm][order])
    nslot = max((len(r) for r in ranks), default=0)
    ranks = [r for r in ranks if len(r) == nslot]
    if ranks:
        r = np.array(ranks)
        print(f"SIMD {s_}: end by wave slot (p50): " + "  ".join(f"{np.percentile(r[:, i], 50):.1f}" for i in range(nslot)))
# occupancy over time: waves alive in 20 slices of the span
edges = np.linspace(0, span, 21)
alive = [(((t0 - start) / 100.0 <= (a + b) / 2) & (e > (a + b) / 2)).sum() for a, b in zip(edges[:-1], edges[1:])]
print("waves alive at the middle of each 5 % slice:", " ".join(str(a) for a in alive))
ph = t[:, 4:16].astype(np.float64)
if ph.sum() > 0:
    pair_names = [f"{c} chain: {n}" for c in ("even", "odd") for n in ("fwd first (loads)", "fwd middle", "fwd last + filter", "inv first", "inv middle", "inv last (+ combine, stores)")]
    names = pair_names if os.environ.get("TRACE_KIND") == "pair" else ["xch wait", "fwd first (HBM loads)", "fwd stage 2", "fwd stage 3", "post-process", "filter + pre-process", "inv first",
             "inv middle", "inv last + own stores", "exchange + stores", "previous stores drained (vmcnt 0)", "-"]
    tot = ph[full].sum(axis=1)
    print(f"phase clocks of the full-length waves (shader cycles per block, p50 over waves; total {np.percentile(tot / nblk[full], 50):.0f}):")
    for i, n in enumerate(names):
        v = ph[full][:, i] / nblk[full]
        print(f"  {n:26s} {np.percentile(v, 50):8.0f}  {100 * np.percentile(v, 50) / np.percentile(tot / nblk[full], 50):5.1f} %")
