"""resampler_amd -- MI355X-native audio resampling engine (host-side mirror of the reference API).

The product is the C-ABI shared library ``libresampler_amd.so`` (HIP kernels for gfx950 + C++ host
runtime, see ``include/resampler_amd.h``).  This module is the thin Python binding the tests and
the bench use; class and method names follow the reference crate (``ResamplerFir::new``,
``new_from_hz``, ``buffer_size_output``, ``resample``, ``delay``, ``reset`` --
src/resampler_fir.rs:252-642; ``ResamplerFft::new``, ``chunk_size_input``, ``chunk_size_output``,
``delay``, ``resample`` -- src/resampler_fft.rs:75-240).

There is no CPU compute path in here: if the library is missing, or there is no HIP device, the
constructors raise.
"""
from __future__ import annotations

import ctypes as C
import enum
import os
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RSMP_AMD_LIB") or os.path.join(_HERE, "libresampler_amd.so")   # override: A/B builds


class ResampleError(Exception):
    """ResampleError (src/error.rs:3-8) plus the argument / device errors of the C ABI."""

    def __init__(self, code: int, message: str):
        super().__init__(f"[{code}] {message}")
        self.code = code


class InvalidInputBufferSize(ResampleError):
    pass


class InvalidOutputBufferSize(ResampleError):
    pass


class SampleRate(enum.IntEnum):
    """enum SampleRate (src/lib.rs:167-188), same order."""
    Hz22050 = 0
    Hz16000 = 1
    Hz32000 = 2
    Hz44100 = 3
    Hz48000 = 4
    Hz88200 = 5
    Hz96000 = 6
    Hz176400 = 7
    Hz192000 = 8
    Hz384000 = 9

    @property
    def hz(self) -> int:
        return lib().rsmp_sample_rate_hz(int(self))


class Latency(enum.IntEnum):
    """enum Latency (src/resampler_fir.rs:139-149)."""
    Sample8 = 0
    Sample16 = 1
    Sample32 = 2
    Sample64 = 3

    def taps(self) -> int:
        return (16, 32, 64, 128)[int(self)]


class Attenuation(enum.IntEnum):
    """enum Attenuation (src/resampler_fir.rs:102-110)."""
    Db60 = 0
    Db90 = 1
    Db120 = 2


class FirKernel(enum.IntEnum):
    Auto = 0
    Generic = 1
    Periodic = 2
    PeriodicVector = 3   # the packed-FMA vector kernel, no matrix cores
    PeriodicF32 = 4      # periodic kernels that keep every product in f32 (never the split-bf16 one)


RSMP_OK = 0
_f32p = C.POINTER(C.c_float)
_szp = C.POINTER(C.c_size_t)


class _Segment(C.Structure):
    _fields_ = [("out_start", C.c_uint32), ("count", C.c_uint32), ("in_base", C.c_int64),
                ("p0", C.c_double), ("inc", C.c_double)]


_lib = None

# Every symbol include/resampler_amd.h declares: (name, restype, argtypes).
_SIGNATURES = [
    ("rsmp_last_error", C.c_char_p, []),
    ("rsmp_device_count", C.c_int, []),
    ("rsmp_version", C.c_char_p, []),
    ("rsmp_sample_rate_hz", C.c_uint32, [C.c_int]),
    ("rsmp_fir_new", C.c_void_p, [C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("rsmp_fir_new_from_hz", C.c_void_p, [C.c_size_t, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int]),
    ("rsmp_fir_free", None, [C.c_void_p]),
    ("rsmp_fir_buffer_size_output", C.c_size_t, [C.c_void_p]),
    ("rsmp_fir_delay", C.c_size_t, [C.c_void_p]),
    ("rsmp_fir_reset", None, [C.c_void_p]),
    ("rsmp_fir_channels", C.c_size_t, [C.c_void_p]),
    ("rsmp_fir_taps", C.c_size_t, [C.c_void_p]),
    ("rsmp_fir_phases", C.c_size_t, [C.c_void_p]),
    ("rsmp_fir_state", None, [C.c_void_p, _szp, _szp, C.POINTER(C.c_double)]),
    ("rsmp_fir_set_kernel", C.c_int, [C.c_void_p, C.c_int]),
    ("rsmp_fir_set_profiling", C.c_int, [C.c_void_p, C.c_int]),
    ("rsmp_fir_last_kernel_ms", C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    ("rsmp_fir_mean_kernel_ms", C.c_int, [C.c_void_p, C.POINTER(C.c_float), _szp]),
    ("rsmp_fir_kernel_variant", C.c_int, [C.c_void_p]),
    ("rsmp_fir_resample", C.c_int, [C.c_void_p, _f32p, C.c_size_t, _f32p, C.c_size_t, _szp, _szp]),
    ("rsmp_fir_resample_device", C.c_int,
     [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, _szp, _szp, C.c_void_p]),
    ("rsmp_fir_bulk_output_bound", C.c_size_t, [C.c_void_p, C.c_size_t, C.c_size_t]),
    ("rsmp_fir_resample_bulk", C.c_int,
     [C.c_void_p, _f32p, C.c_size_t, C.c_size_t, _f32p, C.c_size_t, _szp, _szp, _szp, C.c_size_t, _szp]),
    ("rsmp_fir_resample_bulk_device", C.c_int,
     [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, _szp, _szp, _szp,
      C.c_size_t, _szp, C.c_void_p]),
    ("rsmp_fir_batch_resample_bulk_device", C.c_int,
     [C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(C.c_void_p), _szp, C.c_size_t,
      C.POINTER(C.c_void_p), _szp, _szp, _szp, C.c_void_p]),
    ("rsmp_fir_batch_resample_bulk_device_ex", C.c_int,
     [C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(C.c_void_p), _szp, C.c_size_t,
      C.POINTER(C.c_void_p), _szp, _szp, _szp, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    ("rsmp_fir_batch_reset", None, [C.POINTER(C.c_void_p), C.c_size_t]),
    ("rsmp_fir_lockstep_new", C.c_void_p, [C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t]),
    ("rsmp_fir_lockstep_free", None, [C.c_void_p]),
    ("rsmp_fir_lockstep_size", C.c_size_t, [C.c_void_p]),
    ("rsmp_fir_lockstep_workgroups", C.c_size_t, [C.c_void_p]),
    ("rsmp_fir_lockstep_bind", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _szp]),
    ("rsmp_fir_lockstep_step", C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]),
    ("rsmp_fir_lockstep_counts", C.c_int, [C.c_void_p, _szp, _szp]),
    ("rsmp_fir_lockstep_run", C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]),
    ("rsmp_fir_lockstep_run_counts", C.c_int, [C.c_void_p, _szp, _szp, C.c_size_t]),
    ("rsmp_fir_lockstep_run_bulk", C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]),
    ("rsmp_fir_lockstep_run_slow_calls", C.c_int, [C.c_void_p, _szp]),
    ("rsmp_fir_lockstep_table_rebinds", C.c_int, [C.c_void_p, _szp]),
    ("rsmp_fir_lockstep_status", C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    ("rsmp_fir_lockstep_split_workgroups", C.c_size_t, [C.c_void_p]),
    ("rsmp_fir_lockstep_sync", C.c_int, [C.c_void_p]),
    ("rsmp_fir_lockstep_discard", None, [C.c_void_p]),
    ("rsmp_fir_lockstep_rebind_buffers", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p]),
    ("rsmp_fir_lockstep_sync_totals", C.c_int, [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]),
    ("rsmp_fir_lockstep_in_sync", C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    ("rsmp_fir_batch_distinct_states", C.c_int, [C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(C.c_size_t)]),
    ("rsmp_fir_lockstep_reset", C.c_int, [C.c_void_p]),
    ("rsmp_fir_lockstep_set_profiling", C.c_int, [C.c_void_p, C.c_int]),
    ("rsmp_fir_lockstep_mean_kernel_ms", C.c_int, [C.c_void_p, C.POINTER(C.c_float), _szp]),
    ("rsmp_fir_lockstep_kernel_ms", C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_size_t, _szp]),
    ("rsmp_fir_lockstep_stats", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t]),
    ("rsmp_fir_lockstep_set_drift_policy", C.c_int, [C.c_void_p, C.c_double, C.c_size_t]),
    ("rsmp_design_fir_coeffs", C.c_int, [C.c_uint32, C.c_uint32, C.c_int, C.c_int, _f32p, C.c_size_t]),
    ("rsmp_design_cutoff_kaiser", C.c_double, [C.c_size_t, C.c_double]),
    ("rsmp_fir_plan_new", C.c_void_p, [C.c_uint32, C.c_uint32, C.c_int]),
    ("rsmp_fir_plan_free", None, [C.c_void_p]),
    ("rsmp_fir_plan_reset", None, [C.c_void_p]),
    ("rsmp_fir_plan_state", None, [C.c_void_p, _szp, _szp, C.POINTER(C.c_double)]),
    ("rsmp_fir_plan_call", C.c_int,
     [C.c_void_p, C.c_size_t, C.c_size_t, _szp, _szp, C.POINTER(_Segment), C.c_size_t, _szp]),
    ("rsmp_fir_plan_clone", C.c_void_p, [C.c_void_p]),
    ("rsmp_fir_plan_bulk", C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, _szp, _szp, _szp]),
    ("rsmp_fir_plan_selftest_fast", C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, _szp, _szp, _szp]),
    ("rsmp_fir_seek", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    ("rsmp_interp_output_len", C.c_size_t, [C.c_size_t, C.c_uint32, C.c_uint32, C.c_size_t]),
    ("rsmp_interp_resample", C.c_int,
     [C.c_int, C.c_size_t, C.c_uint32, C.c_uint32, _f32p, C.c_size_t, _f32p, C.c_size_t, _szp]),
    ("rsmp_interp_resample_device", C.c_int,
     [C.c_int, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, _szp, C.c_void_p]),
    ("rsmp_pcm_to_stereo_f32_device", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]),
    ("rsmp_device_stream_copy", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("rsmp_fft_new", C.c_void_p, [C.c_size_t, C.c_int, C.c_int, C.c_int]),
    ("rsmp_fft_free", None, [C.c_void_p]),
    ("rsmp_fft_chunk_size_input", C.c_size_t, [C.c_void_p]),
    ("rsmp_fft_chunk_size_output", C.c_size_t, [C.c_void_p]),
    ("rsmp_fft_delay", C.c_size_t, [C.c_void_p]),
    ("rsmp_fft_channels", C.c_size_t, [C.c_void_p]),
    ("rsmp_fft_set_profiling", C.c_int, [C.c_void_p, C.c_int]),
    ("rsmp_fft_last_kernel_ms", C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    ("rsmp_fft_resample", C.c_int, [C.c_void_p, _f32p, C.c_size_t, _f32p, C.c_size_t]),
    ("rsmp_fft_resample_device", C.c_int,
     [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("rsmp_fft_resample_bulk", C.c_int, [C.c_void_p, _f32p, C.c_size_t, _f32p, C.c_size_t, C.c_size_t]),
    ("rsmp_fft_resample_bulk_device", C.c_int,
     [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]),
    ("rsmp_fft_batch_resample_bulk_device", C.c_int,
     [C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _szp, C.c_void_p]),
    ("rsmp_fir_batch_resample_bulk_pcm_device", C.c_int,
     [C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(C.c_void_p), C.c_int, _szp, C.c_size_t, C.POINTER(C.c_void_p), _szp, _szp, _szp, C.c_void_p]),
    ("rsmp_fft_batch_resample_bulk_pcm_device", C.c_int,
     [C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p), _szp, C.c_void_p]),
    ("rsmp_fft_plan_sizes", C.c_int,
     [C.c_uint32, C.c_uint32, _szp, _szp, C.POINTER(C.c_int), _szp, C.POINTER(C.c_int), _szp, C.c_size_t]),
]


def declared_symbols() -> Sequence[str]:
    return [s[0] for s in _SIGNATURES]


def lib() -> C.CDLL:
    """Loads libresampler_amd.so (built by __graft_entry__.build / make -C resampler_amd/csrc)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no Python/CPU fallback for the HIP path)")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64;
    # if /opt/rocm's copy gets loaded first, torch later loads a second runtime that finds no
    # GPU.  Importing torch first makes the dynamic loader bind this library to torch's copy
    # (same SONAME).  Without torch installed the system runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    for name, restype, argtypes in _SIGNATURES:
        fn = getattr(L, name, None)
        if fn is None and os.environ.get("RSMP_AMD_LIB"):
            continue   # (an A/B library built from an older commit: tools/ab_headline.sh)
        if fn is None:
            raise ImportError(f"{LIB_PATH} does not export {name}: rebuild it")
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = L
    return L


def last_error() -> str:
    return lib().rsmp_last_error().decode()


def device_count() -> int:
    return lib().rsmp_device_count()


STREAM_LEGACY = 1   # RSMP_STREAM_LEGACY: the legacy default stream as a `stream` argument (None / 0 = the handle's own stream)


def torch_stream(stream=None) -> int:
    """The `stream` argument that makes a device entry point run on a torch stream (default: torch's current
    one).  torch reports its default stream as handle 0, which at the C ABI means "the handle's own non-blocking
    stream" -- work there is not ordered against torch's; this maps it to RSMP_STREAM_LEGACY."""
    import torch
    s = torch.cuda.current_stream() if stream is None else stream
    return int(s.cuda_stream) or STREAM_LEGACY


def _check(rc: int) -> None:
    if rc == RSMP_OK:
        return
    msg = last_error()
    if rc == 1:
        raise InvalidInputBufferSize(rc, msg)
    if rc == 2:
        raise InvalidOutputBufferSize(rc, msg)
    raise ResampleError(rc, msg)


def _np_f32(a) -> np.ndarray:
    a = np.asarray(a)
    if a.dtype != np.float32 or not a.flags["C_CONTIGUOUS"]:
        a = np.ascontiguousarray(a, np.float32)
    return a


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(_f32p)


def _dev_ptr(t) -> int:
    """Device pointer of a torch CUDA tensor (float32, contiguous)."""
    assert t.is_cuda and t.is_contiguous() and str(t.dtype) == "torch.float32"
    return t.data_ptr()


class ResamplerFir:
    """GPU-backed ResamplerFir (src/resampler_fir.rs:179-643)."""

    def __init__(self, channels: int, input_rate, output_rate, latency: Latency = Latency.Sample64,
                 attenuation: Attenuation = Attenuation.Db120, device: int = 0, *, _from_hz=False):
        L = lib()
        if _from_hz:
            h = L.rsmp_fir_new_from_hz(channels, int(input_rate), int(output_rate), int(latency),
                                       int(attenuation), device)
        else:
            h = L.rsmp_fir_new(channels, int(SampleRate(input_rate)), int(SampleRate(output_rate)),
                               int(latency), int(attenuation), device)
        if not h:
            raise ResampleError(3, last_error())
        self._h = C.c_void_p(h)
        self.device = device

    # ResamplerFir::new / new_from_hz -------------------------------------------------------------
    @classmethod
    def new(cls, channels, input_rate: SampleRate, output_rate: SampleRate,
            latency: Latency = Latency.Sample64, attenuation: Attenuation = Attenuation.Db120,
            device: int = 0) -> "ResamplerFir":
        return cls(channels, input_rate, output_rate, latency, attenuation, device)

    @classmethod
    def new_from_hz(cls, channels, input_rate_hz: int, output_rate_hz: int,
                    latency: Latency = Latency.Sample64, attenuation: Attenuation = Attenuation.Db120,
                    device: int = 0) -> "ResamplerFir":
        if input_rate_hz < 0 or output_rate_hz < 0:
            raise ValueError("sample rates are u32")
        return cls(channels, input_rate_hz, output_rate_hz, latency, attenuation, device, _from_hz=True)

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().rsmp_fir_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __repr__(self) -> str:  # fmt::Debug, resampler_fir.rs:203-211
        return f"ResamplerFir {{ channels: {self.channels}, taps: {self.taps}, phases: {self.phases}, .. }}"

    @property
    def channels(self) -> int:
        return lib().rsmp_fir_channels(self._h)

    @property
    def taps(self) -> int:
        return lib().rsmp_fir_taps(self._h)

    @property
    def phases(self) -> int:
        return lib().rsmp_fir_phases(self._h)

    def buffer_size_output(self) -> int:
        return lib().rsmp_fir_buffer_size_output(self._h)

    def delay(self) -> int:
        return lib().rsmp_fir_delay(self._h)

    def reset(self) -> None:
        lib().rsmp_fir_reset(self._h)

    def state(self):
        """(read_position, available_frames, position) -- resampler_fir.rs:189-192."""
        rp, av, pos = C.c_size_t(), C.c_size_t(), C.c_double()
        lib().rsmp_fir_state(self._h, C.byref(rp), C.byref(av), C.byref(pos))
        return rp.value, av.value, pos.value

    def seek(self, plan: "FirPlan", history, stream: Optional[int] = None) -> None:
        """Starts this resampler where ``plan`` stands: its state, and as buffered frames the end of
        ``history`` -- the input preceding the point (numpy array or CUDA tensor).  See sharding.fir_time_shards."""
        if isinstance(history, np.ndarray):
            h = _np_f32(history)
            _check(lib().rsmp_fir_seek(self._h, plan._h, _ptr(h), h.size, 0, C.c_void_p(stream or 0)))
        else:
            _check(lib().rsmp_fir_seek(self._h, plan._h, C.c_void_p(_dev_ptr(history) if history.numel() else 0),
                                       history.numel(), 1, C.c_void_p(stream or 0)))

    def set_kernel(self, kernel: FirKernel) -> None:
        _check(lib().rsmp_fir_set_kernel(self._h, int(kernel)))

    def set_profiling(self, enable: bool) -> None:
        _check(lib().rsmp_fir_set_profiling(self._h, 1 if enable else 0))

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        _check(lib().rsmp_fir_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def mean_kernel_ms(self) -> Tuple[float, int]:
        ms, n = C.c_float(), C.c_size_t()
        _check(lib().rsmp_fir_mean_kernel_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def kernel_variant(self) -> int:
        """0 generic, 1 periodic vector, 2 periodic vector (double-buffered), 3 periodic f32 matrix-core,
        4 periodic split matrix-core with three bf16 planes, 5 with two fp16 planes (default)."""
        return int(lib().rsmp_fir_kernel_variant(self._h))

    # ResamplerFir::resample (host slices) --------------------------------------------------------
    def resample(self, input, output: np.ndarray) -> Tuple[int, int]:
        """Returns (consumed, produced) in f32 values; raises ResampleError like Err(..)."""
        inp = _np_f32(input)
        assert output.dtype == np.float32 and output.flags["C_CONTIGUOUS"]
        c, p = C.c_size_t(), C.c_size_t()
        _check(lib().rsmp_fir_resample(self._h, _ptr(inp), inp.size, _ptr(output), output.size,
                                       C.byref(c), C.byref(p)))
        return c.value, p.value

    def resample_device(self, d_in, d_out, stream: Optional[int] = None) -> Tuple[int, int]:
        """Same call on HBM-resident torch tensors (async on `stream`, a hipStream_t value)."""
        c, p = C.c_size_t(), C.c_size_t()
        _check(lib().rsmp_fir_resample_device(self._h, _dev_ptr(d_in), d_in.numel(), _dev_ptr(d_out),
                                              d_out.numel(), C.byref(c), C.byref(p),
                                              C.c_void_p(stream or 0)))
        return c.value, p.value

    # Bulk: the CLI driver loop (resample/src/main.rs:226-254) in one launch ------------------------
    def bulk_output_bound(self, in_len: int, chunk_len: int = 512) -> int:
        return lib().rsmp_fir_bulk_output_bound(self._h, in_len, chunk_len)

    def resample_bulk(self, input, chunk_len: int = 512, want_calls: bool = False):
        """Returns (output[:produced], consumed) or (output, consumed, calls[n,2])."""
        inp = _np_f32(input)
        cap = self.bulk_output_bound(inp.size, chunk_len)
        out = np.empty(cap, np.float32)
        c, p, nc = C.c_size_t(), C.c_size_t(), C.c_size_t()
        max_calls = (inp.size // max(1, chunk_len) + 2) if want_calls else 0
        calls = np.zeros(2 * max(1, max_calls), np.uintp)
        _check(lib().rsmp_fir_resample_bulk(self._h, _ptr(inp), inp.size, chunk_len, _ptr(out), cap,
                                            C.byref(c), C.byref(p),
                                            calls.ctypes.data_as(_szp) if want_calls else None,
                                            max_calls, C.byref(nc)))
        if want_calls:
            k = min(nc.value, max_calls)
            return out[:p.value], c.value, calls[:2 * k].reshape(k, 2).astype(np.int64)
        return out[:p.value], c.value

    def resample_bulk_device(self, d_in, d_out, chunk_len: int = 512, stream: Optional[int] = None,
                             in_len: Optional[int] = None) -> Tuple[int, int]:
        c, p, nc = C.c_size_t(), C.c_size_t(), C.c_size_t()
        n_in = d_in.numel() if in_len is None else in_len
        _check(lib().rsmp_fir_resample_bulk_device(self._h, _dev_ptr(d_in), n_in, chunk_len,
                                                   _dev_ptr(d_out), d_out.numel(), C.byref(c),
                                                   C.byref(p), None, 0, C.byref(nc),
                                                   C.c_void_p(stream or 0)))
        return c.value, p.value


class FirBatch:
    """N independent ResamplerFir instances on one device processed in one launch per step
    (rsmp_fir_batch_resample_bulk_device) -- the unit sharded across GPUs."""

    def __init__(self, resamplers: Sequence[ResamplerFir]):
        self.resamplers = list(resamplers)
        n = len(self.resamplers)
        self._handles = (C.c_void_p * n)(*[r._h for r in self.resamplers])
        self._in = (C.c_void_p * n)()
        self._out = (C.c_void_p * n)()
        self._in_lens = (C.c_size_t * n)()
        self._out_caps = (C.c_size_t * n)()
        self._consumed = (C.c_size_t * n)()
        self._produced = (C.c_size_t * n)()

    def bind(self, d_ins, d_outs) -> None:
        """Binds one input and one output tensor per stream (kept until the next bind)."""
        self._keep = (list(d_ins), list(d_outs))
        for i, (a, b) in enumerate(zip(d_ins, d_outs)):
            self._in[i] = _dev_ptr(a)
            self._out[i] = _dev_ptr(b)
            self._in_lens[i] = a.numel()
            self._out_caps[i] = b.numel()

    def reset(self) -> None:
        lib().rsmp_fir_batch_reset(self._handles, len(self.resamplers))

    # A bulk launch plans every DISTINCT state among its streams on the host (streams in one state share a plan): 2-3 ms for 64
    # streams in 64 states around a 0.3 ms kernel.  Such a batch goes through the device planner instead, behind the same entry
    # (rsmp_fir_batch_resample_bulk_device_ex: a lock-step batch over the same handles, kept by the library from launch to launch,
    # the states written back into the handles before the call returns -- which therefore returns when the launch is THROUGH).
    # `device_planner`: None = the library decides (at least 16 different states, ...), False = never, True = whenever it can;
    # `planned_on_device`: which planner the last launch had.
    kDevicePlanStates = 16
    device_planner: Optional[bool] = None
    planned_on_device = False

    def resample_bulk_device(self, chunk_len: int = 512, stream: Optional[int] = None):
        n = len(self.resamplers)
        took = C.c_int()
        mode = -1 if self.device_planner is None else (1 if self.device_planner else 0)
        _check(lib().rsmp_fir_batch_resample_bulk_device_ex(
            self._handles, n, self._in, self._in_lens, chunk_len, self._out, self._out_caps,
            self._consumed, self._produced, C.c_void_p(stream or 0), mode, C.byref(took)))
        self.planned_on_device = bool(took.value)
        # zero-copy views (valid until the next call): converting 2 x n ctypes words to Python
        # ints costs more than the launch for batches of a thousand streams
        return (np.ctypeslib.as_array(self._consumed), np.ctypeslib.as_array(self._produced))


    def resample_bulk_pcm_device(self, d_pcms, bits: int, d_outs, chunk_len: int = 512, stream: Optional[int] = None):
        """The bulk driver loop over two-channel WAV samples as they are in the file (rsmp_fir_batch_resample_bulk_pcm_device):
        d_pcms = uint8 tensors of little-endian PCM, `bits` per sample; converted where the kernels read their input."""
        n = len(self.resamplers)
        pin, pout = (C.c_void_p * n)(), (C.c_void_p * n)()
        lens, caps = (C.c_size_t * n)(), (C.c_size_t * n)()
        for i, (a, b) in enumerate(zip(d_pcms, d_outs)):
            pin[i], pout[i] = a.data_ptr(), _dev_ptr(b)
            lens[i], caps[i] = a.numel() // (bits // 8), b.numel()
        self._keep_pcm = (list(d_pcms), list(d_outs))
        _check(lib().rsmp_fir_batch_resample_bulk_pcm_device(self._handles, n, pin, bits, lens, chunk_len, pout, caps,
                                                             self._consumed, self._produced, C.c_void_p(stream or 0)))
        return (np.ctypeslib.as_array(self._consumed), np.ctypeslib.as_array(self._produced))


class FirLockstep:
    """A fixed set of ResamplerFir streams stepped together, one resample() call per stream and step
    (rsmp_fir_lockstep_*): the streams' state lives in HBM, a step is one kernel launch and no
    per-stream host work.  BASELINE config 4's shape."""

    def __init__(self, resamplers: Sequence[ResamplerFir], max_step_frames: int = 512):
        self.resamplers = list(resamplers)
        n = len(self.resamplers)
        self._handles = (C.c_void_p * n)(*[r._h for r in self.resamplers])
        h = lib().rsmp_fir_lockstep_new(self._handles, n, max_step_frames)
        if not h:
            raise ResampleError(3, last_error())
        self._h = C.c_void_p(h)
        self._in = (C.c_void_p * n)()
        self._out = (C.c_void_p * n)()
        self._out_caps = (C.c_size_t * n)()
        self._consumed = (C.c_size_t * n)()
        self._produced = (C.c_size_t * n)()
        self._last_run = 0

    discard_on_close = False   # True: close() leaves the handles' states alone (FirBatch's own batch: written back after every launch)

    def close(self) -> None:
        if getattr(self, "_h", None):
            (lib().rsmp_fir_lockstep_discard if self.discard_on_close else lib().rsmp_fir_lockstep_free)(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def workgroups(self) -> int:
        return lib().rsmp_fir_lockstep_workgroups(self._h)

    def split_workgroups(self) -> int:
        """Workgroups per step that run on the fp16 matrix cores with split operands (two-channel streams)."""
        return lib().rsmp_fir_lockstep_split_workgroups(self._h)

    def bind(self, d_ins, d_outs) -> None:
        self._keep = (list(d_ins), list(d_outs))
        for i, (a, b) in enumerate(zip(d_ins, d_outs)):
            self._in[i] = _dev_ptr(a)
            self._out[i] = _dev_ptr(b)
            self._out_caps[i] = b.numel()
        _check(lib().rsmp_fir_lockstep_bind(self._h, self._in, self._out, self._out_caps))

    def bind_caps(self, d_ins, d_outs, out_caps: Sequence[int]) -> None:
        """Like bind, with an explicit per-step output capacity (append mode: the tensors are longer)."""
        self._keep = (list(d_ins), list(d_outs))
        for i, (a, b, k) in enumerate(zip(d_ins, d_outs, out_caps)):
            self._in[i] = _dev_ptr(a)
            self._out[i] = _dev_ptr(b)
            self._out_caps[i] = k
        _check(lib().rsmp_fir_lockstep_bind(self._h, self._in, self._out, self._out_caps))

    def rebind(self, d_ins, d_outs, stream: Optional[int] = None) -> None:
        """New buffers, same capacities (rsmp_fir_lockstep_rebind_buffers): a run planned ahead that starts at the front of the
        output survives."""
        self._keep = (list(d_ins), list(d_outs))
        for i, (a, b) in enumerate(zip(d_ins, d_outs)):
            self._in[i] = _dev_ptr(a)
            self._out[i] = _dev_ptr(b)
        _check(lib().rsmp_fir_lockstep_rebind_buffers(self._h, self._in, self._out, C.c_void_p(stream or 0)))

    def step(self, in_frames: int, in_offset_frames: int = 0, append: bool = False,
             stream: Optional[int] = None, d_in_frames=None) -> None:
        ptr = None
        if d_in_frames is not None:
            assert d_in_frames.is_cuda and str(d_in_frames.dtype) == "torch.int32"
            ptr = C.c_void_p(d_in_frames.data_ptr())
        _check(lib().rsmp_fir_lockstep_step(self._h, in_frames, in_offset_frames, ptr, 1 if append else 0,
                                            C.c_void_p(stream or 0)))
        self._last_run = 0

    def run(self, k_steps: int, in_frames: int, in_offset_frames: int = 0, append: bool = True,
            stream: Optional[int] = None) -> None:
        """k_steps consecutive steps in one go (rsmp_fir_lockstep_run): planned on the device, computed by the bulk
        kernels; the calls' outputs follow each other in the output buffers."""
        _check(lib().rsmp_fir_lockstep_run(self._h, k_steps, in_frames, in_offset_frames, 1 if append else 0,
                                           C.c_void_p(stream or 0)))
        self._last_run = k_steps

    def run_bulk(self, total_frames: int, chunk_frames: int, in_offset_frames: int = 0, append: bool = False,
                 stream: Optional[int] = None) -> None:
        """A whole buffer per stream in calls of `chunk_frames` frames, the last one shorter (the reference's driver
        loop, resample/src/main.rs:226-254), planned on the device: rsmp_fir_lockstep_run_bulk.  run_counts() has the
        equal calls' counts, counts() the last call's when total_frames is no multiple of chunk_frames."""
        _check(lib().rsmp_fir_lockstep_run_bulk(self._h, total_frames, chunk_frames, in_offset_frames, 1 if append else 0,
                                                C.c_void_p(stream or 0)))
        self._last_run = total_frames // chunk_frames

    def run_counts(self):
        """(consumed, produced) of every call of the last run: two int64 arrays [k_steps][streams], in f32 values."""
        k, n = self._last_run, len(self.resamplers)
        if k == 0:   # (no run yet, or a step since: nothing to report -- the C side says the same with an error)
            return np.zeros((0, n), np.int64), np.zeros((0, n), np.int64)
        cons, prod = (C.c_size_t * (k * n))(), (C.c_size_t * (k * n))()
        _check(lib().rsmp_fir_lockstep_run_counts(self._h, cons, prod, k))
        return (np.ctypeslib.as_array(cons).astype(np.int64).reshape(k, n),
                np.ctypeslib.as_array(prod).astype(np.int64).reshape(k, n))

    def table_rebinds(self) -> int:
        """Times a class of the batch's streams got new class tables because the f64 position drift had moved on (diagnostic)."""
        v = C.c_size_t()
        _check(lib().rsmp_fir_lockstep_table_rebinds(self._h, C.byref(v)))
        return v.value

    STAT_NAMES = ("table_rebinds", "plan_ahead_hits", "plan_ahead_misses", "late_table_polls", "table_waits",
                  "plan_stream_probes", "has_plan_stream", "drift_classes")

    def stats(self) -> dict:
        """Diagnostic counters of the batch (rsmp_fir_lockstep_stats)."""
        v = (C.c_uint64 * len(self.STAT_NAMES))()
        _check(lib().rsmp_fir_lockstep_stats(self._h, v, len(self.STAT_NAMES)))
        return {k: int(x) for k, x in zip(self.STAT_NAMES, v)}

    def set_drift_policy(self, tolerance_frames: float, check_frames: int) -> None:
        """How closely the class tables follow the streams' f64 drift (rsmp_fir_lockstep_set_drift_policy)."""
        _check(lib().rsmp_fir_lockstep_set_drift_policy(self._h, tolerance_frames, check_frames))

    def kernel_ms(self, max_launches: int = 256) -> np.ndarray:
        """Device time of each of the last profiled launches (steps or runs), oldest first, in ms."""
        ms, n = (C.c_float * max_launches)(), C.c_size_t()
        _check(lib().rsmp_fir_lockstep_kernel_ms(self._h, ms, max_launches, C.byref(n)))
        return np.ctypeslib.as_array(ms)[:n.value].astype(np.float64).copy()

    def run_slow_calls(self) -> int:
        """Calls of the last run that the device planner's fast path declined (diagnostic)."""
        v = C.c_size_t()
        _check(lib().rsmp_fir_lockstep_run_slow_calls(self._h, C.byref(v)))
        return v.value

    def counts(self):
        """(consumed, produced) of the last step per stream, in f32 values (waits for the step)."""
        _check(lib().rsmp_fir_lockstep_counts(self._h, self._consumed, self._produced))
        return (np.ctypeslib.as_array(self._consumed).astype(np.int64),
                np.ctypeslib.as_array(self._produced).astype(np.int64))

    def status(self) -> np.ndarray:
        st = (C.c_uint32 * len(self.resamplers))()
        _check(lib().rsmp_fir_lockstep_status(self._h, st))
        return np.ctypeslib.as_array(st).copy()

    def sync(self) -> None:
        _check(lib().rsmp_fir_lockstep_sync(self._h))

    def set_profiling(self, enable: bool) -> None:
        _check(lib().rsmp_fir_lockstep_set_profiling(self._h, 1 if enable else 0))

    def mean_kernel_ms(self) -> Tuple[float, int]:
        ms, n = C.c_float(), C.c_size_t()
        _check(lib().rsmp_fir_lockstep_mean_kernel_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def reset(self) -> None:
        _check(lib().rsmp_fir_lockstep_reset(self._h))


class ResamplerFft:
    """GPU-backed ResamplerFft (src/resampler_fft.rs:43-240)."""

    def __init__(self, channels: int, sample_rate_input: SampleRate, sample_rate_output: SampleRate,
                 device: int = 0):
        h = lib().rsmp_fft_new(channels, int(SampleRate(sample_rate_input)),
                               int(SampleRate(sample_rate_output)), device)
        if not h:
            raise ResampleError(3, last_error())
        self._h = C.c_void_p(h)
        self.device = device

    @classmethod
    def new(cls, channels, sample_rate_input: SampleRate, sample_rate_output: SampleRate,
            device: int = 0) -> "ResamplerFft":
        return cls(channels, sample_rate_input, sample_rate_output, device)

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().rsmp_fft_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __repr__(self) -> str:  # fmt::Debug, resampler_fft.rs:56-66
        ch = self.channels
        return (f"ResamplerFft {{ channels: {ch}, chunk_size_input: {self.chunk_size_input()}, "
                f"chunk_size_output: {self.chunk_size_output()}, fft_size_input: "
                f"{self.chunk_size_input() // ch}, fft_size_output: {self.chunk_size_output() // ch}, .. }}")

    @property
    def channels(self) -> int:
        return lib().rsmp_fft_channels(self._h)

    def chunk_size_input(self) -> int:
        return lib().rsmp_fft_chunk_size_input(self._h)

    def chunk_size_output(self) -> int:
        return lib().rsmp_fft_chunk_size_output(self._h)

    def delay(self) -> int:
        return lib().rsmp_fft_delay(self._h)

    def set_profiling(self, enable: bool) -> None:
        _check(lib().rsmp_fft_set_profiling(self._h, 1 if enable else 0))

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        _check(lib().rsmp_fft_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def resample(self, input, output: np.ndarray) -> None:
        """One chunk; raises InvalidInputBufferSize / InvalidOutputBufferSize like Err(..)."""
        inp = _np_f32(input)
        assert output.dtype == np.float32 and output.flags["C_CONTIGUOUS"]
        _check(lib().rsmp_fft_resample(self._h, _ptr(inp), inp.size, _ptr(output), output.size))

    def resample_bulk(self, input, n_chunks: int) -> np.ndarray:
        inp = _np_f32(input)
        out = np.empty(n_chunks * self.chunk_size_output(), np.float32)
        _check(lib().rsmp_fft_resample_bulk(self._h, _ptr(inp), inp.size, _ptr(out), out.size, n_chunks))
        return out

    def resample_batch(self, input) -> np.ndarray:
        """The CLI's whole-file driver `resample_batch` (resample/src/main.rs:256-313): complete
        chunks, then the zero-padded partial chunk, trimmed to ceil(len * out / in) values."""
        inp = _np_f32(input)
        n_in, n_out = self.chunk_size_input(), self.chunk_size_output()
        total = -(-inp.size // n_in)
        padded = np.zeros(total * n_in, np.float32)
        padded[:inp.size] = inp
        out = self.resample_bulk(padded, total) if total else np.zeros(0, np.float32)
        expected = int(np.ceil(float(inp.size) * float(n_out) / float(n_in)))
        return out[:expected]

    def resample_device(self, d_in, d_out, stream: Optional[int] = None) -> None:
        _check(lib().rsmp_fft_resample_device(self._h, _dev_ptr(d_in), d_in.numel(), _dev_ptr(d_out),
                                              d_out.numel(), C.c_void_p(stream or 0)))

    def resample_bulk_device(self, d_in, d_out, n_chunks: int, stream: Optional[int] = None) -> None:
        _check(lib().rsmp_fft_resample_bulk_device(self._h, _dev_ptr(d_in), d_in.numel(),
                                                   _dev_ptr(d_out), d_out.numel(), n_chunks,
                                                   C.c_void_p(stream or 0)))


class FftBatch:
    """N ResamplerFft instances with one rate pair on one device, one launch per step."""

    def __init__(self, resamplers: Sequence[ResamplerFft]):
        self.resamplers = list(resamplers)
        n = len(self.resamplers)
        self._handles = (C.c_void_p * n)(*[r._h for r in self.resamplers])
        self._in = (C.c_void_p * n)()
        self._out = (C.c_void_p * n)()
        self._chunks = (C.c_size_t * n)()

    def bind(self, d_ins, d_outs, n_chunks: Sequence[int]) -> None:
        self._keep = (list(d_ins), list(d_outs))
        for i, (a, b, k) in enumerate(zip(d_ins, d_outs, n_chunks)):
            r = self.resamplers[i]
            assert a.numel() >= k * r.chunk_size_input() and b.numel() >= k * r.chunk_size_output()
            self._in[i] = _dev_ptr(a)
            self._out[i] = _dev_ptr(b)
            self._chunks[i] = k

    def resample_bulk_device(self, stream: Optional[int] = None) -> None:
        _check(lib().rsmp_fft_batch_resample_bulk_device(self._handles, len(self.resamplers), self._in,
                                                         self._out, self._chunks, C.c_void_p(stream or 0)))

    def resample_bulk_pcm_device(self, d_pcms, bits: int, d_outs, n_chunks: Sequence[int], stream: Optional[int] = None) -> None:
        """The same over WAV samples as they are in the file (rsmp_fft_batch_resample_bulk_pcm_device): d_pcms = uint8
        tensors of little-endian PCM, `bits` per sample, two channels a frame; converted in the kernel's first load."""
        n = len(self.resamplers)
        pin, pout, ch = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_size_t * n)()
        for i, (a, b, k) in enumerate(zip(d_pcms, d_outs, n_chunks)):
            r = self.resamplers[i]
            assert a.numel() >= k * r.chunk_size_input() * (bits // 8) and b.numel() >= k * r.chunk_size_output()
            pin[i], pout[i], ch[i] = a.data_ptr(), _dev_ptr(b), k
        self._keep_pcm = (list(d_pcms), list(d_outs))
        _check(lib().rsmp_fft_batch_resample_bulk_pcm_device(self._handles, n, pin, bits, pout, ch, C.c_void_p(stream or 0)))


class InterpolationMode(enum.IntEnum):
    """enum InterpolationMode (resample/src/interpolation_resampler.rs:4-10)."""
    Linear = 0
    Hermite = 1


class InterpolationResampler:
    """The CLI's comparison interpolators (resample/src/interpolation_resampler.rs:12-127) on the GPU."""

    def __init__(self, channels: int, input_rate: SampleRate, output_rate: SampleRate, mode: InterpolationMode):
        self.channels = channels
        self.in_hz = SampleRate(input_rate).hz
        self.out_hz = SampleRate(output_rate).hz
        self.mode = InterpolationMode(mode)

    def resample(self, input) -> np.ndarray:
        inp = _np_f32(input)
        cap = lib().rsmp_interp_output_len(self.channels, self.in_hz, self.out_hz, inp.size)
        out = np.empty(max(cap, 1), np.float32)
        p = C.c_size_t()
        _check(lib().rsmp_interp_resample(int(self.mode), self.channels, self.in_hz, self.out_hz, _ptr(inp),
                                          inp.size, _ptr(out), cap, C.byref(p)))
        return out[:p.value]


def pcm_to_stereo_f32_device(d_pcm, bits: int, channels: int, d_out, stream: Optional[int] = None) -> None:
    """WAV sample conversion (resample/src/main.rs:128-156) on HBM-resident torch tensors: d_pcm = uint8
    bytes of little-endian samples, d_out = float32 (mono input: two values per sample)."""
    n = d_pcm.numel() // (bits // 8)
    assert d_out.numel() >= n * (2 if channels == 1 else 1)
    _check(lib().rsmp_pcm_to_stereo_f32_device(C.c_void_p(d_pcm.data_ptr()), bits, channels, n,
                                               C.c_void_p(d_out.data_ptr()), C.c_void_p(stream or 0)))


def fft_plan_sizes(input_rate_hz: int, output_rate_hz: int):
    """(fft_size_input, fft_size_output, forward stage radices, inverse stage radices)."""
    fi, fo, nf, ni = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
    a = (C.c_int * 16)()
    b = (C.c_int * 16)()
    _check(lib().rsmp_fft_plan_sizes(input_rate_hz, output_rate_hz, C.byref(fi), C.byref(fo), a,
                                     C.byref(nf), b, C.byref(ni), 16))
    return fi.value, fo.value, list(a[:nf.value]), list(b[:ni.value])


# ---- host-only helpers (no GPU needed) -----------------------------------------------------------
def design_fir_coeffs(input_rate_hz: int, output_rate_hz: int, latency: Latency,
                      attenuation: Attenuation) -> np.ndarray:
    taps = Latency(latency).taps()
    out = np.empty((1024, taps), np.float32)
    _check(lib().rsmp_design_fir_coeffs(input_rate_hz, output_rate_hz, int(latency), int(attenuation),
                                        _ptr(out), out.size))
    return out


def design_cutoff_kaiser(sample_count: int, beta: float) -> float:
    return lib().rsmp_design_cutoff_kaiser(sample_count, beta)


class FirPlan:
    """Host mirror of the ResamplerFir state machine (counts and exact positions, no samples)."""

    def __init__(self, input_rate_hz: int, output_rate_hz: int, latency: Latency = Latency.Sample64):
        h = lib().rsmp_fir_plan_new(input_rate_hz, output_rate_hz, int(latency))
        if not h:
            raise ResampleError(3, last_error())
        self._h = C.c_void_p(h)

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:   # (module globals are gone at interpreter exit)
            lib().rsmp_fir_plan_free(self._h)
            self._h = None

    def reset(self) -> None:
        lib().rsmp_fir_plan_reset(self._h)

    def state(self):
        rp, av, pos = C.c_size_t(), C.c_size_t(), C.c_double()
        lib().rsmp_fir_plan_state(self._h, C.byref(rp), C.byref(av), C.byref(pos))
        return rp.value, av.value, pos.value

    def clone(self) -> "FirPlan":
        c = FirPlan.__new__(FirPlan)
        c._h = C.c_void_p(lib().rsmp_fir_plan_clone(self._h))
        return c

    def bulk(self, in_frames: int, chunk_frames: int, max_calls: int = 0):
        """The CLI driver loop (resample/src/main.rs:226-254) on the plan alone: calls of ``chunk_frames``
        until ``in_frames`` are used up or ``max_calls`` calls were made -> (accepted, produced, calls)."""
        a, p, n = C.c_size_t(), C.c_size_t(), C.c_size_t()
        _check(lib().rsmp_fir_plan_bulk(self._h, in_frames, chunk_frames, max_calls, C.byref(a), C.byref(p),
                                        C.byref(n)))
        return a.value, p.value, n.value

    def call(self, input_frames: int, output_capacity_frames: int, want_segments: bool = False):
        """One resample() call in frames -> (accepted, produced[, segments])."""
        a, p, ns = C.c_size_t(), C.c_size_t(), C.c_size_t()
        max_segs = 8192 if want_segments else 0
        segs = (_Segment * max(1, max_segs))()
        _check(lib().rsmp_fir_plan_call(self._h, input_frames, output_capacity_frames, C.byref(a),
                                        C.byref(p), segs if want_segments else None, max_segs,
                                        C.byref(ns)))
        if want_segments:
            return a.value, p.value, [(s.out_start, s.count, s.in_base, s.p0, s.inc)
                                      for s in segs[:ns.value]]
        return a.value, p.value


def device_stream_copy(d_src, d_dst, stream: Optional[int] = None) -> None:
    """rsmp_device_stream_copy: a plain 16-bytes-per-lane copy between two device tensors (measurement aid)."""
    assert d_src.is_cuda and d_dst.is_cuda and d_src.numel() == d_dst.numel()
    _check(lib().rsmp_device_stream_copy(C.c_void_p(d_src.data_ptr()), C.c_void_p(d_dst.data_ptr()), d_src.numel(),
                                         C.c_void_p(stream or 0)))
