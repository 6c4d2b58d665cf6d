"""The split FIR kernel keeps HBM loads in flight across its loop's back edge by issuing them from inline asm -- which
hides from the compiler that their destination registers are not valid yet.  Round 3 shipped a three-plane build whose
register allocator copied such registers at the loop's latch (wrong results, found late): tools/asm_inflight_lint.py
walks the ISA of every instantiation (a forward data flow over basic blocks: registers written by an asm load until the
`s_waitcnt vmcnt` that lands them) and reports any read in between.  hipcc cross-compiles without a GPU (~40 s)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_no_instruction_reads_a_register_an_asm_load_still_has_in_flight():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_inflight_lint.py"),
                          os.path.join(ROOT, "resampler_amd", "csrc", "fir_split.hip")],
                         capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-2000:]
    assert " 0 reads of a register in flight" in out.stdout, out.stdout[-2000:]
    kernels = int(out.stdout.split(":")[1].split("kernels")[0])
    assert kernels >= 30, out.stdout   # (every instantiation was looked at)
