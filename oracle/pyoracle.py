"""ctypes front-end of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package (resampler_amd) never does.  See oracle/oracle.h for the pinning status.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
# The same sources built as BASELINE.md section 2 prescribes for the timed CPU baseline (-O3 -mavx2 -mfma, still
# -ffp-contract=off: floor / trunc / conversions inline, loops vectorised, no operation fused or reordered).
# tests/test_oracle_native.py checks that it returns the portable build's values bit for bit.
_NATIVE_PATH = os.path.join(_HERE, "liboracle_native.so")
NATIVE_BUILD_FLAGS = "gcc -O3 -mavx2 -mfma -ffp-contract=off -fno-fast-math"
PORTABLE_BUILD_FLAGS = "gcc -O2 -ffp-contract=off -fno-fast-math (AVX+FMA only inside convolve_interp_avx_fma)"

WINDOW_PERIODIC = 0
WINDOW_SYMMETRIC = 1
CONVOLVE_SCALAR = 0
CONVOLVE_AVX_FMA = 1
CONVOLVE_AVX512 = 2   # fir/avx512.rs: what the reference's dispatch takes first (timing / differential test; not the parity leaf)


def build(force: bool = False) -> str:
    """Compile oracle/*.c into oracle/liboracle.so with gcc (make)."""
    if force:
        subprocess.check_call(["make", "-C", _HERE, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_libs = {}
_native = False


def cpu_has_avx2_fma() -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("flags"):
                    fl = ln.split()
                    return "avx2" in fl and "fma" in fl
    except OSError:
        pass
    return False


def use_native(flag: bool) -> bool:
    """Objects created from now on come from liboracle_native.so (False: back to liboracle.so).  Returns whether
    the native build is in use (it needs AVX2 + FMA on this CPU).  An object keeps the library that made it."""
    global _native
    _native = bool(flag) and cpu_has_avx2_fma() and (os.path.exists(_NATIVE_PATH) or _try_build())
    return _native


def _try_build() -> bool:
    try:
        build()
    except Exception:
        return False
    return os.path.exists(_NATIVE_PATH)


def build_flags() -> str:
    return NATIVE_BUILD_FLAGS if _native else PORTABLE_BUILD_FLAGS


def lib() -> C.CDLL:
    path = _NATIVE_PATH if _native else _LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    f32p = C.POINTER(C.c_float)
    szp = C.POINTER(C.c_size_t)
    L.orc_bessel_i0.restype = C.c_double
    L.orc_bessel_i0.argtypes = [C.c_double]
    L.orc_make_kaiser_window.restype = None
    L.orc_make_kaiser_window.argtypes = [C.c_size_t, C.c_double, C.c_int, f32p]
    L.orc_calculate_cutoff_kaiser.restype = C.c_double
    L.orc_calculate_cutoff_kaiser.argtypes = [C.c_size_t, C.c_double]
    L.orc_make_sincs_for_kaiser.restype = None
    L.orc_make_sincs_for_kaiser.argtypes = [C.c_size_t, C.c_size_t, C.c_float, C.c_double, C.c_int, f32p]
    for name in ("orc_convolve_interp_scalar", "orc_convolve_interp_avx_fma", "orc_convolve_interp_avx512"):
        fn = getattr(L, name)
        fn.restype = C.c_float
        fn.argtypes = [f32p, f32p, f32p, C.c_float, C.c_size_t]
    L.orc_have_avx_fma.restype = C.c_int
    L.orc_have_avx512f.restype = C.c_int
    L.orc_fir_bench_calls.restype = C.c_ulonglong
    L.orc_fir_bench_calls.argtypes = [C.c_void_p, f32p, C.c_size_t, f32p, C.c_size_t, C.c_size_t]
    L.orc_fir_new.restype = C.c_void_p
    L.orc_fir_new.argtypes = [C.c_size_t, C.c_uint32, C.c_uint32, C.c_size_t, C.c_int, C.c_int]
    L.orc_fir_free.argtypes = [C.c_void_p]
    L.orc_fir_buffer_size_output.restype = C.c_size_t
    L.orc_fir_buffer_size_output.argtypes = [C.c_void_p]
    L.orc_fir_delay.restype = C.c_size_t
    L.orc_fir_delay.argtypes = [C.c_void_p]
    L.orc_fir_reset.argtypes = [C.c_void_p]
    L.orc_fir_resample.restype = C.c_int
    L.orc_fir_resample.argtypes = [C.c_void_p, f32p, C.c_size_t, f32p, C.c_size_t, szp, szp]
    L.orc_fir_coeffs.restype = f32p
    L.orc_fir_coeffs.argtypes = [C.c_void_p]
    L.orc_fir_ratio.restype = C.c_double
    L.orc_fir_ratio.argtypes = [C.c_void_p]
    L.orc_fir_state.argtypes = [C.c_void_p, szp, szp, C.POINTER(C.c_double)]
    L.orc_fir_seek.restype = C.c_int
    L.orc_fir_seek.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_double, C.POINTER(C.c_float), C.c_size_t]
    if hasattr(L, "orc_fir_skip_calls"):
        L.orc_fir_skip_calls.restype = C.c_ulonglong
        L.orc_fir_skip_calls.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_ulonglong)]
    L.orc_fir_resample_all.restype = C.c_size_t
    L.orc_fir_resample_all.argtypes = [C.c_void_p, f32p, C.c_size_t, C.c_size_t, f32p, C.c_size_t,
                                       szp, C.c_size_t, szp]
    if hasattr(L, "orc_fft_new"):
        ip = C.POINTER(C.c_int)
        L.orc_fft_plan.restype = C.c_int
        L.orc_fft_plan.argtypes = [C.c_uint32, C.c_uint32, szp, szp, ip, szp, ip, szp, C.c_int]
        L.orc_optimize_factors.restype = C.c_size_t
        L.orc_optimize_factors.argtypes = [ip, C.c_size_t]
        L.orc_rfft_new.restype = C.c_void_p
        L.orc_rfft_new.argtypes = [ip, C.c_size_t, C.c_int]
        L.orc_rfft_free.argtypes = [C.c_void_p]
        L.orc_rfft_len.restype = C.c_size_t
        L.orc_rfft_len.argtypes = [C.c_void_p]
        L.orc_rfft_stage_factors.restype = C.c_size_t
        L.orc_rfft_stage_factors.argtypes = [C.c_void_p, ip]
        L.orc_rfft_forward.argtypes = [C.c_void_p, f32p, f32p]
        L.orc_rfft_inverse.argtypes = [C.c_void_p, f32p, f32p]
        L.orc_fft_new.restype = C.c_void_p
        L.orc_fft_new.argtypes = [C.c_size_t, C.c_uint32, C.c_uint32]
        L.orc_fft_free.argtypes = [C.c_void_p]
        for name in ("orc_fft_chunk_size_input", "orc_fft_chunk_size_output", "orc_fft_delay"):
            fn = getattr(L, name)
            fn.restype = C.c_size_t
            fn.argtypes = [C.c_void_p]
        L.orc_fft_resample.restype = C.c_int
        L.orc_fft_resample.argtypes = [C.c_void_p, f32p, C.c_size_t, f32p, C.c_size_t]
        L.orc_fft_filter_spectrum.restype = f32p
        L.orc_fft_filter_spectrum.argtypes = [C.c_void_p, szp]
    L.orc_fft_resample_all.restype = C.c_size_t
    L.orc_fft_resample_all.argtypes = [C.c_void_p, f32p, C.c_size_t, f32p, C.c_size_t]
    for name in ("orc_interp_linear", "orc_interp_hermite"):
        fn = getattr(L, name)
        fn.restype = C.c_size_t
        fn.argtypes = [C.c_size_t, C.c_uint32, C.c_uint32, f32p, C.c_size_t, f32p, C.c_size_t]
    L.orc_pcm_to_stereo_f32.restype = C.c_size_t
    L.orc_pcm_to_stereo_f32.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_size_t, f32p]
    _libs[path] = L
    return L


def _f32p(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


# ---- window.rs -----------------------------------------------------------------------------
def bessel_i0(x: float) -> float:
    return lib().orc_bessel_i0(x)


def make_kaiser_window(n: int, beta: float, window_type: int) -> np.ndarray:
    out = np.empty(n, np.float32)
    lib().orc_make_kaiser_window(n, beta, window_type, _f32p(out))
    return out


def calculate_cutoff_kaiser(sample_count: int, beta: float) -> float:
    return lib().orc_calculate_cutoff_kaiser(sample_count, beta)


def make_sincs_for_kaiser(sample_count, factor, f_cutoff, beta, window_type) -> np.ndarray:
    out = np.empty((factor, sample_count), np.float32)
    lib().orc_make_sincs_for_kaiser(sample_count, factor, f_cutoff, beta, window_type, _f32p(out))
    return out


def convolve_interp(x, c1, c2, frac, kind=CONVOLVE_SCALAR) -> float:
    x = np.ascontiguousarray(x, np.float32)
    taps = len(c1)
    # AVX path needs 64-byte aligned rows (fir/avx.rs:19-20).
    buf = np.zeros(2 * taps + 32, np.float32)
    off = (-buf.ctypes.data % 64) // 4
    a1 = buf[off:off + taps]
    o2 = off + ((taps + 15) // 16) * 16
    a2 = buf[o2:o2 + taps]
    a1[:] = c1
    a2[:] = c2
    fn = (lib().orc_convolve_interp_avx512 if kind == CONVOLVE_AVX512 else
          lib().orc_convolve_interp_avx_fma if kind == CONVOLVE_AVX_FMA else lib().orc_convolve_interp_scalar)
    return float(fn(_f32p(x), a1.ctypes.data_as(C.POINTER(C.c_float)),
                    a2.ctypes.data_as(C.POINTER(C.c_float)), frac, taps))


def have_avx_fma() -> bool:
    return bool(lib().orc_have_avx_fma())


def have_avx512f() -> bool:
    return bool(lib().orc_have_avx512f())


# ---- resampler_fir.rs ----------------------------------------------------------------------
class OracleFir:
    """Mirror of ResamplerFir (resampler_fir.rs:179-643) on the oracle."""

    def __init__(self, channels, in_hz, out_hz, taps=128, attenuation_db=120, kind=CONVOLVE_SCALAR):
        self._L = lib()
        self._h = self._L.orc_fir_new(channels, in_hz, out_hz, taps, attenuation_db, kind)
        if not self._h:
            raise ValueError("invalid ResamplerFir arguments")
        self.channels = channels
        self.taps = taps

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_fir_free(self._h)
            self._h = None

    def buffer_size_output(self) -> int:
        return self._L.orc_fir_buffer_size_output(self._h)

    def delay(self) -> int:
        return self._L.orc_fir_delay(self._h)

    def reset(self) -> None:
        self._L.orc_fir_reset(self._h)

    @property
    def ratio(self) -> float:
        return self._L.orc_fir_ratio(self._h)

    def coeffs(self) -> np.ndarray:
        p = self._L.orc_fir_coeffs(self._h)
        return np.ctypeslib.as_array(p, shape=(1024, self.taps)).copy()

    def state(self):
        rp, av, pos = C.c_size_t(), C.c_size_t(), C.c_double()
        self._L.orc_fir_state(self._h, C.byref(rp), C.byref(av), C.byref(pos))
        return rp.value, av.value, pos.value

    def seek(self, state, history: np.ndarray) -> None:
        """Test support: (read_position, available_frames, position) + the input preceding the point."""
        h = np.ascontiguousarray(history, np.float32)
        if self._L.orc_fir_seek(self._h, state[0], state[1], state[2], _f32p(h), h.size) != 0:
            raise ValueError("orc_fir_seek: history shorter than the buffered frames")

    def skip_calls(self, calls: int, in_frames: int):
        """`calls` resample() calls of `in_frames` frames each, control flow only (the f64 position recurrence output by
        output, no samples): returns (frames accepted, frames produced).  The ring's contents are stale afterwards:
        `seek(self.state(), history)` before the next real call."""
        c = C.c_ulonglong()
        p = self._L.orc_fir_skip_calls(self._h, calls, in_frames, C.byref(c))
        return c.value, p

    def resample(self, inp: np.ndarray, out: np.ndarray):
        """Returns (status, consumed, produced); status 0/1/2 as error.rs:3-8."""
        inp = np.ascontiguousarray(inp, np.float32)
        c, p = C.c_size_t(), C.c_size_t()
        rc = self._L.orc_fir_resample(self._h, _f32p(inp), inp.size, _f32p(out), out.size,
                                    C.byref(c), C.byref(p))
        return rc, c.value, p.value

    def resample_all(self, inp: np.ndarray, chunk_len: int = 512, max_calls: int | None = None):
        """main.rs:226-254 driver loop.  Returns (output, calls[n,2])."""
        inp = np.ascontiguousarray(inp, np.float32)
        ratio = self.ratio
        cap = int(inp.size / ratio) + 4 * self.buffer_size_output()
        out = np.empty(cap, np.float32)
        if max_calls is None:
            max_calls = inp.size // max(1, min(chunk_len, 64)) + 16
        calls = np.zeros(2 * max_calls, np.uintp)
        nc = C.c_size_t()
        n = self._L.orc_fir_resample_all(self._h, _f32p(inp), inp.size, chunk_len, _f32p(out), cap,
                                       calls.ctypes.data_as(C.POINTER(C.c_size_t)), max_calls,
                                       C.byref(nc))
        k = min(nc.value, max_calls)
        return out[:n].copy(), calls[:2 * k].reshape(k, 2).astype(np.int64)


    def bench_calls(self, inp: np.ndarray, out: np.ndarray, iterations: int) -> int:
        """The criterion bench's inner loop (benches/benchmark_resampler_fir.rs:50-89): `iterations` resample() calls on the
        same input; returns the values produced in total."""
        return int(self._L.orc_fir_bench_calls(self._h, _f32p(inp), inp.size, _f32p(out), out.size, iterations))

    def resample_all_into(self, inp: np.ndarray, chunk_len: int, out: np.ndarray) -> int:
        """The same driver loop into a caller-owned buffer (no allocation: timing loops on many threads)."""
        nc = C.c_size_t()
        return self._L.orc_fir_resample_all(self._h, _f32p(inp), inp.size, chunk_len, _f32p(out), out.size,
                                            None, 0, C.byref(nc))


# ---- fft -----------------------------------------------------------------------------------
def fft_plan(in_hz: int, out_hz: int, scale: bool = True):
    fi, fo = C.c_size_t(), C.c_size_t()
    a = (C.c_int * 32)()
    b = (C.c_int * 32)()
    na, nb = C.c_size_t(), C.c_size_t()
    rc = lib().orc_fft_plan(in_hz, out_hz, C.byref(fi), C.byref(fo), a, C.byref(na), b, C.byref(nb),
                            1 if scale else 0)
    if rc != 0:
        raise ValueError("unsupported sample rate")
    return fi.value, fo.value, list(a[:na.value]), list(b[:nb.value])


def optimize_factors(factors):
    arr = (C.c_int * max(1, len(factors)))(*factors)
    n = lib().orc_optimize_factors(arr, len(factors))
    return list(arr[:n])


class OracleRfft:
    """RadixFFT<Forward|Inverse> (radix_fft.rs:105-713) on the oracle."""

    def __init__(self, factors, inverse=False, simd=False):
        arr = (C.c_int * len(factors))(*factors)
        self._L = lib()
        self._L.orc_rfft_new_simd.restype = C.c_void_p
        self._L.orc_rfft_new_simd.argtypes = [C.POINTER(C.c_int), C.c_size_t, C.c_int, C.c_int]
        self._h = self._L.orc_rfft_new_simd(arr, len(factors), 1 if inverse else 0, 1 if simd else 0)
        if not self._h:
            raise ValueError("bad factors")
        self.n = self._L.orc_rfft_len(self._h)
        self.inverse = inverse

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_rfft_free(self._h)
            self._h = None

    def stage_factors(self):
        arr = (C.c_int * 32)()
        n = self._L.orc_rfft_stage_factors(self._h, arr)
        return list(arr[:n])

    def forward(self, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float32)
        assert x.size == self.n and not self.inverse
        out = np.empty(2 * (self.n // 2 + 1), np.float32)
        self._L.orc_rfft_forward(self._h, _f32p(x), _f32p(out))
        return out.view(np.complex64)

    def inverse_transform(self, X: np.ndarray) -> np.ndarray:
        X = np.ascontiguousarray(X, np.complex64)
        assert X.size == self.n // 2 + 1 and self.inverse
        out = np.empty(self.n, np.float32)
        self._L.orc_rfft_inverse(self._h, _f32p(X.view(np.float32)), _f32p(out))
        return out


class OracleFft:
    """Mirror of ResamplerFft (resampler_fft.rs:43-240) on the oracle."""

    def __init__(self, channels, in_hz, out_hz, simd=False):
        """simd: the reference's AVX + FMA butterflies and real <-> complex passes (oracle/fft_avx.c) instead of the scalar
        specs (use_native(True) selects the -O3 -mavx2 -mfma build of the same sources: the timed CPU baseline)."""
        self._L = lib()
        self._L.orc_fft_new_simd.restype = C.c_void_p
        self._L.orc_fft_new_simd.argtypes = [C.c_size_t, C.c_uint32, C.c_uint32, C.c_int]
        self._h = self._L.orc_fft_new_simd(channels, in_hz, out_hz, 1 if simd else 0)
        if not self._h:
            raise ValueError("invalid ResamplerFft arguments")
        self.channels = channels

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_fft_free(self._h)
            self._h = None

    def chunk_size_input(self):
        return self._L.orc_fft_chunk_size_input(self._h)

    def chunk_size_output(self):
        return self._L.orc_fft_chunk_size_output(self._h)

    def delay(self):
        return self._L.orc_fft_delay(self._h)

    def filter_spectrum(self) -> np.ndarray:
        n = C.c_size_t()
        p = self._L.orc_fft_filter_spectrum(self._h, C.byref(n))
        return np.ctypeslib.as_array(p, shape=(2 * n.value,)).copy().view(np.complex64)

    def resample(self, inp: np.ndarray, out: np.ndarray) -> int:
        inp = np.ascontiguousarray(inp, np.float32)
        return self._L.orc_fft_resample(self._h, _f32p(inp), inp.size, _f32p(out), out.size)

    def resample_all(self, inp: np.ndarray) -> np.ndarray:
        """resample_batch (resample/src/main.rs:256-313): whole buffer, last chunk zero padded, output trimmed."""
        inp = np.ascontiguousarray(inp, np.float32)
        ci, co = self.chunk_size_input(), self.chunk_size_output()
        out = np.zeros(((inp.size + ci - 1) // ci) * co, np.float32)
        n = self._L.orc_fft_resample_all(self._h, _f32p(inp), inp.size, _f32p(out), out.size)
        if n == 0 and inp.size:
            raise ValueError("orc_fft_resample_all failed")
        return out[:n]

    def resample_all_into(self, inp: np.ndarray, out: np.ndarray) -> int:
        return self._L.orc_fft_resample_all(self._h, _f32p(inp), inp.size, _f32p(out), out.size)


# ---- resample/src: interpolators and WAV sample conversion ----------------------------------------
def interpolate(mode: str, channels: int, in_hz: int, out_hz: int, x: np.ndarray) -> np.ndarray:
    """InterpolationResampler::resample (interpolation_resampler.rs:41-126); mode 'linear' | 'hermite'."""
    x = np.ascontiguousarray(x, np.float32)
    frames = x.size // channels
    cap = (int(np.ceil(frames * (out_hz / in_hz))) + 2) * channels
    out = np.zeros(cap, np.float32)
    fn = lib().orc_interp_linear if mode == "linear" else lib().orc_interp_hermite
    n = fn(channels, in_hz, out_hz, _f32p(x), x.size, _f32p(out), cap)
    return out[:n * channels].copy()


def pcm_to_stereo_f32(pcm: bytes, bits: int, channels: int) -> np.ndarray:
    """main.rs:128-156."""
    n = len(pcm) // (bits // 8)
    out = np.zeros(n * (2 if channels == 1 else 1), np.float32)
    w = lib().orc_pcm_to_stereo_f32(pcm, bits, channels, n, _f32p(out))
    return out[:w]
