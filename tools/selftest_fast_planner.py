"""tools/selftest_fast_planner.py [--verbose] -- the run planner's fast path (fir_mirror_fast.h) against the plain state
machine on the host (rsmp_fir_plan_selftest_fast): all ordered pairs of twelve rates x five (latency, call size, calls,
run length) shapes; prints the calls compared, the mismatches (must be 0) and the share of calls the fast path declined."""
import ctypes as C
import itertools
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import resampler_amd as ra


def selftest(in_hz, out_hz, lat, frames, calls, runlen, prefeed=0):
    L = ra.lib()
    L.rsmp_fir_plan_selftest_fast.restype = C.c_int
    L.rsmp_fir_plan_selftest_fast.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.rsmp_fir_plan_new.restype = C.c_void_p
    L.rsmp_fir_plan_new.argtypes = [C.c_uint32, C.c_uint32, C.c_int]
    L.rsmp_fir_plan_free.argtypes = [C.c_void_p]
    p = L.rsmp_fir_plan_new(in_hz, out_hz, lat)
    bad, slow, lean = C.c_size_t(), C.c_size_t(), C.c_size_t()
    if prefeed:   # a different state to start from
        assert L.rsmp_fir_plan_selftest_fast(p, prefeed, 3, 1, C.byref(bad), C.byref(slow), C.byref(lean)) == 0 and bad.value == 0
    assert L.rsmp_fir_plan_selftest_fast(p, frames, calls, runlen, C.byref(bad), C.byref(slow), C.byref(lean)) == 0
    L.rsmp_fir_plan_free(p)
    return bad.value, slow.value, lean.value


RATES = [8000, 11025, 16000, 22050, 32000, 44100, 48000, 88200, 96000, 176400, 192000, 384000]
SHAPES = [(3, 512, 3000, 256), (1, 200, 1500, 16), (0, 1000, 800, 64), (2, 37, 2500, 100), (3, 3900, 300, 7)]

if __name__ == "__main__":
    verbose = "--verbose" in sys.argv
    tot_bad = tot_slow = tot = tot_lean = 0
    t0 = time.time()
    for i, o_ in itertools.permutations(RATES, 2):
        for lat, frames, calls, runlen in SHAPES:
            bad, slow, lean = selftest(i, o_, lat, frames, calls, runlen, prefeed=(i // 100) % 300)
            tot_bad += bad
            tot_slow += slow
            tot_lean += lean
            tot += calls
            if bad or (verbose and slow * 20 > calls):
                print(f"{i} -> {o_} latency {lat} frames {frames}: {calls} calls, {bad} mismatches, {slow} slow")
    print(f"calls {tot} mismatches {tot_bad} slow {tot_slow} ({100.0 * tot_slow / tot:.1f} %) unchecked chain {tot_lean} "
          f"({100.0 * tot_lean / tot:.1f} %) {time.time() - t0:.1f} s")
