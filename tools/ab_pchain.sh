#!/bin/bash
# tools/ab_pchain.sh [reps] -- GPU box: config 4 (1024 / 128 / 64 streams, 256 steps per launch) and the distinct-states bulk launch with the
# planner's chain in parallel per chunk of 64 calls (default) and, through RSMP_LS_PCHAIN=0, call by call as in round 5.
REPS=${1:-2}
for rep in $(seq $REPS); do
  for n in 1024 128 64; do
    for k in 1 0; do
      RSMP_DEBUG=1 RSMP_LS_PCHAIN=$k timeout -k 5 200 python bench.py --config c4 --c4-streams $n --steps 16384 --warmup 512 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('streams %4d parallel_chain=$k  us/step %.3f' % ($n, d['ms_per_step']*1e3))"
    done
  done
done
