#!/bin/bash
# tools/c4_timelines.sh TAG -- kernel timelines (rocprofv3 --kernel-trace, tools/kt_timeline.py) of config 4's run of 256 calls at
# 1024 and 128 streams, planned ahead and not: where a run's time goes at the batch and at the shard an 8-GPU rank gets.
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:?run through gpurun}
OUT="$R/gpurun_out/$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for n in 1024 128; do
  for ahead in 1 0; do
    d="$OUT/kt_${n}_a$ahead"
    RSMP_DEBUG=1 RSMP_LS_AHEAD=$ahead timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d "$d" -- python3 "$R/tools/run_probe.py" $n 256 > /dev/null 2> "$d.err"
    python3 "$R/tools/kt_timeline.py" "$(find "$d" -name '*kernel_trace.csv' | head -1)" 24 > "$OUT/c4_run_timeline_${n}_ahead$ahead.txt" 2>/dev/null
  done
done
