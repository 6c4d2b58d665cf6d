#!/bin/bash
# tools/ch_probe.sh "CH IN OUT" ... -- GPU box: tools/ch_probe.py under `rocprofv3 --kernel-trace --stats` for each shape and
# each library (the shipping one and every libresampler_amd_exp*.so): the launch's kernels and what each takes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ $# -eq 0 ] && set -- "3 44100 48000" "1 48000 44100" "4 44100 48000"
for cfg in "$@"; do
  for lib in $R/resampler_amd/libresampler_amd.so $R/resampler_amd/libresampler_amd_exp*.so; do
    [ -f "$lib" ] || continue
    rm -rf /tmp/prof_ch
    echo "== $(basename $lib)"
    RSMP_AMD_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ch -- python3 $R/tools/ch_probe.py $cfg 10 2>&1 | grep " ch "
    f=$(find /tmp/prof_ch -name "*kernel_stats.csv" | head -1)
    python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:3]:
    print("   %-70s calls %5s avg %10.1f us  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
  done
done
