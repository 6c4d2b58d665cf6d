"""Every switch of the library that can change WHAT is computed (read only under RSMP_DEBUG=1, once per process) run in a
process of its own against the oracle -- so that no alternative path ships untested (round 3: RSMP_FIR_SPLIT_PLANES=3
returned wrong samples for a whole round because nothing ran it).  The diagnostic switches (RSMP_FIR_DEBUG / _TRACE /
_WTRACE / _VERBOSE, RSMP_LS_TRACE, RSMP_FIR_MFMA_DBG) change timing or print, not results, and are not here."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FIR = [("RSMP_FIR_SPLIT_LONG", "0", "2 96000 44100 2 48000 96000"),
       ("RSMP_FIR_SPLIT_WIDE", "0", "8 44100 48000 1 48000 44100"),
       ("RSMP_FIR_SPLIT_QUADS", "0", "8 96000 44100 16 96000 44100"),
       ("RSMP_FIR_SPLIT_PLANES", "3", "2 44100 48000 2 48000 44100"),
       ("RSMP_FIR_MFMA", "0", "2 44100 48000 4 48000 44100"),
       ("RSMP_FIR_MFMA", "1", "2 44100 48000 2 96000 44100"),
       ("RSMP_FIR_MFMA_RING", "1", "2 44100 48000")]
FFT = [("RSMP_FFT_WAVE_NOC2", "1", "44100 48000 2 25 44100 48000 4 20"),
       ("RSMP_FFT_WAVE_WIDE", "3", "22050 96000 2 12 88200 96000 2 12"),
       ("RSMP_FFT_WAVE", "0", "44100 48000 2 25"),
       ("RSMP_FFT_PAIR", "0", "44100 48000 2 25"),        # two-channel streams on the wave-per-channel kernel
       ("RSMP_FFT_PAIR_SHARE", "0.5", "48000 44100 2 40"),  # equal runs for a SIMD's old and young wave
       ("RSMP_FFT_GENERIC", "1", "44100 48000 2 25")]


def _child(script, args, knob, value):
    env = dict(os.environ, RSMP_DEBUG="1", PYTHONPATH=ROOT)
    env[knob] = value
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script)] + args.split(), env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    return p.stdout


@pytest.mark.parametrize("knob,value,args", FIR)
def test_fir_switch_in_a_child_process(knob, value, args):
    out = _child("split_geo_check.py", args, knob, value)
    lines = [ln for ln in out.splitlines() if " rms " in ln]
    assert len(lines) >= 3, out
    for ln in lines:
        assert "counts_equal True" in ln and ln.rstrip().endswith("bad 0"), ln
        assert float(re.search(r"rms ([0-9.e+-]+)", ln).group(1)) <= 1e-6, ln


@pytest.mark.parametrize("knob,value,args", FFT)
def test_fft_switch_in_a_child_process(knob, value, args):
    out = _child("fft_pair_check.py", args, knob, value)
    lines = [ln for ln in out.splitlines() if "worst block rms" in ln]
    assert lines, out
    for ln in lines:
        assert "bad blocks []" in ln, ln
        assert float(re.search(r"worst block rms ([0-9.e+-]+)", ln).group(1)) <= 1e-6, ln


def test_lockstep_exact_f32_switch_in_a_child_process():
    env = dict(os.environ, RSMP_DEBUG="1", RSMP_LS_EXACT="1", PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_fir_lockstep_gpu.py"), "-q", "-m", "gpu",
                        "-k", "(c4_shape or channel_counts or different_states) and not runs_split and not mixed_batches", "-p", "no:cacheprovider"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:]


@pytest.mark.parametrize("knob,value", [("RSMP_FIR_SPLIT_MULTI", "0"), ("RSMP_LS_AHEAD", "0"), ("RSMP_LS_PCHAIN", "0"), ("RSMP_LS_COMMIT_ON_PLAN", "0"),
                                        ("RSMP_FIR_SPLIT_ALL", "0"), ("RSMP_LS_PACK", "1"), ("RSMP_LS_FUSE_COMMIT", "0"), ("RSMP_LS_EARLY_WAIT", "0"),
                                        ("RSMP_LS_STOP_EVENT", "0"), ("RSMP_LS_LAZY_DONE", "0"), ("RSMP_LS_ITEMS_AHEAD", "0")])
def test_lockstep_run_switches_in_a_child_process(knob, value):
    """The run of several calls with one launch per rate pair (no multi-job launch) / planned on the caller's stream only / the
    planner's chain call by call (round 5's) instead of a chunk of calls per wave in parallel / the states committed on the
    caller's stream / one split launch per kernel build instead of one for all / one planner wave per workgroup / round 6's
    changes between two split launches one by one (the commit a launch of its own, the wait for the planner in front of the
    next run, events recorded by packets, the "computed" event always recorded, the item tables built in front of the kernel)."""
    env = dict(os.environ, RSMP_DEBUG="1", PYTHONPATH=ROOT)
    env[knob] = value
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_fir_lockstep_run_gpu.py"), "-q", "-m", "gpu",
                        "-k", "planned_ahead or k_calls_equals or interleave or append or different_states or same_span or old_stream or big_batch", "-p", "no:cacheprovider"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:]
    assert " passed" in p.stdout, p.stdout[-1000:]


def test_long_generic_launches_on_the_latency_kernel_in_a_child_process():
    """RSMP_FIR_GENERIC_BULK=0: ratios without a short period keep fir_generic_kernel for launches of any length."""
    env = dict(os.environ, RSMP_DEBUG="1", RSMP_FIR_GENERIC_BULK="0", PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_fir_gpu.py"), "-q", "-m", "gpu",
                        "-k", "without_a_short_period", "-p", "no:cacheprovider"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:]
    assert " passed" in p.stdout, p.stdout[-1000:]


def test_without_rsmp_debug_no_switch_is_read():
    """RSMP_FIR_SPLIT_PLANES=3 alone (no RSMP_DEBUG) must leave the default kernel in place: variant 5, not 4."""
    env = dict(os.environ, RSMP_FIR_SPLIT_PLANES="3", PYTHONPATH=ROOT)
    env.pop("RSMP_DEBUG", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "split_geo_check.py"), "2", "44100", "48000"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "variant 5" in p.stdout and "variant 4" not in p.stdout, p.stdout
