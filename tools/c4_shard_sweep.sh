#!/bin/bash
# tools/c4_shard_sweep.sh [k ...] -- config 4 on one GPU at the shard sizes an 8-GPU run hands a rank (1024 / 512 / 256 / 128
# streams), for k lock-step calls per launch (default 1 16 64 256): t(1024) / t(128) is the strong-scaling figure of the
# 8-GPU run.
KS=${@:-1 16 64 256}
for k in $KS; do
  for n in 1024 512 256 128; do
    steps=$(( k > 64 ? 64 * k : 2048 ))
    python3 bench.py --config c4 --c4-streams $n --c4-k $k --steps $steps --warmup $(( k > 16 ? k : 16 )) --spinup-seconds 0.5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('k %4d  streams %5d  us/step %8.3f  events %8.3f  value %9.1f %s  frac %.4f' % ($k, $n, d['ms_per_step'] * 1e3, r['kernel_ms'] * 1e3, d['value'], d['unit'], r['frac']))"
  done
done
