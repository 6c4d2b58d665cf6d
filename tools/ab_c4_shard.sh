#!/bin/bash
# tools/ab_c4_shard.sh [reps] -- GPU box: config 4 at 1024 streams and at the shard sizes of an 8-GPU run (128, 64), 256 steps per
# launch: round 6's run path (the three kernel builds in one launch, the commit on the plan stream, the planner's waves packed four
# to a CU) against round 5's through the debug knobs, interleaved.
REPS=${1:-3}
for rep in $(seq $REPS); do
  for n in 1024 128 64; do
    for cfg in "1 1 0" "0 0 1"; do
      set -- $cfg
      RSMP_DEBUG=1 RSMP_LS_COMMIT_ON_PLAN=$1 RSMP_FIR_SPLIT_ALL=$2 RSMP_LS_PACK=$3 timeout -k 5 200 python bench.py --config c4 --c4-streams $n --steps 32 --warmup 4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('streams %4d  %s  us/step %.3f' % ($n, 'round 6' if $1 else 'round 5 (knobs)', d['ms_per_step']*1e3))"
    done
  done
done
