#!/bin/bash
# tools/c4_shard_sweep.sh [extra bench args] -- config 4 on one GPU at the shard sizes an 8-GPU run hands a rank
# (1024 / 512 / 256 / 128 streams): t(1024) / t(128) is the strong-scaling figure of the 8-GPU run.
for n in 1024 512 256 128; do
  for rep in 1 2; do
    python3 bench.py --config c4 --c4-streams $n --steps 256 --warmup 16 --spinup-seconds 0.5 "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('streams %5d  ms/step %.5f  kernel %.5f  value %.1f %s  frac %.4f  wgs %s  %s' % ($n, d['ms_per_step'], r['kernel_ms'], d['value'], d['unit'], r['frac'], d['config'].get('workgroups_this_rank'), d['config'].get('steps_per_launch','')))"
  done
done
