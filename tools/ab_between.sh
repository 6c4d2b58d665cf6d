#!/bin/bash
# tools/ab_between.sh [reps] -- GPU box: what stands between two split launches of config 4's run (1024 / 128 streams, 256 calls), switch by
# switch (each =0 with the others at their defaults): the commit fused into K1 (RSMP_LS_FUSE_COMMIT), the wait for the planner behind the
# split launch (RSMP_LS_EARLY_WAIT), events completed by the launches themselves (RSMP_LS_STOP_EVENT), no "computed" event for a big
# batch (RSMP_LS_LAZY_DONE), the item tables built on the plan stream (RSMP_LS_ITEMS_AHEAD).  64 runs per figure.
REPS=${1:-2}
for rep in $(seq $REPS); do
  for n in 1024 128; do
    for knob in NONE RSMP_LS_FUSE_COMMIT RSMP_LS_EARLY_WAIT RSMP_LS_STOP_EVENT RSMP_LS_LAZY_DONE RSMP_LS_ITEMS_AHEAD ALL; do
      if [ $knob = ALL ]; then
        line=$(RSMP_DEBUG=1 RSMP_LS_FUSE_COMMIT=0 RSMP_LS_EARLY_WAIT=0 RSMP_LS_STOP_EVENT=0 RSMP_LS_LAZY_DONE=0 RSMP_LS_ITEMS_AHEAD=0 PROBE_STREAM=1 PROBE_RUNS=64 PROBE_REPS=3 python tools/run_probe.py $n 256 2>/dev/null | tail -1)
      else
        line=$(env RSMP_DEBUG=1 $knob=0 PROBE_STREAM=1 PROBE_RUNS=64 PROBE_REPS=3 python tools/run_probe.py $n 256 2>/dev/null | tail -1)
      fi
      echo "off: $knob  $line"
    done
  done
done
