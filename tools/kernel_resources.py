#!/usr/bin/env python3
"""tools/kernel_resources.py <file.hip> [extra hipcc flags] -- registers / spills / scratch of every kernel in a HIP
source of resampler_amd/csrc (device-only compile with -Rpass-analysis=kernel-resource-usage; one line per kernel)."""
import os
import re
import subprocess
import sys

here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "resampler_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950",
       "--cuda-device-only", "-c", sys.argv[1], "-o", "/tmp/kr_%d.co" % os.getpid(),
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
out = subprocess.run(cmd, cwd=here, capture_output=True, text=True).stderr
rows, cur = [], {}
for ln in out.splitlines():
    if "error" in ln and "remark" not in ln:
        print(ln)
    m = re.search(r"remark: \s*([A-Za-z ]+?)(?: \[bytes/lane\]| \[bytes/block\])?: (\S+)", ln)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    else:
        cur[k] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = name.replace("(anonymous namespace)::", "").replace("rsmp::", "").replace("void ", "")
    name = re.sub(r"\((rsmp::)?FirStreamDesc.*|\(.*\)$", "", name)
    g = r.get
    print("%-58s vgpr %4s agpr %3s sgpr %4s sspill %3s vspill %3s scratch %4s occ %s lds %s" % (
        name, g("VGPRs"), g("AGPRs"), g("TotalSGPRs"), g("SGPRs Spill"), g("VGPRs Spill"), g("ScratchSize"), g("Occupancy"), g("LDS Size")))
try:
    os.remove("/tmp/kr_%d.co" % os.getpid())
except OSError:
    pass
