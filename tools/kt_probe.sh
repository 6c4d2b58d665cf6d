#!/bin/bash
# tools/kt_probe.sh <name> <python script + args ...> -- GPU box: the command under `rocprofv3 --kernel-trace --stats`, the ten
# kernels that take most of its GPU time (calls, average, share), and the command's own last stdout line.
NAME=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt_$NAME
echo "== $NAME: $*"
( cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$NAME -- python3 "$@" 2>/dev/null | tail -1 | cut -c1-400 )
f=$(find /tmp/kt_$NAME -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:10]:
    print("   %-86s calls %6s avg %9.1f us  %6s %%" % (r["Name"][:86], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
