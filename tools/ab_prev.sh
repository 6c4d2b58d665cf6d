#!/bin/bash
# tools/ab_prev.sh [bench args] -- the bench line for resampler_amd/libresampler_amd_expprev.so (an earlier commit's build:
# git worktree add build/wt_prev <commit>; make -C build/wt_prev/resampler_amd/csrc ../libresampler_amd.so; cp) and this
# tree's library, three rounds interleaved inside one lease.
for rep in 1 2 3; do
  for lib in resampler_amd/libresampler_amd_expprev.so resampler_amd/libresampler_amd.so; do
    [ -f "$lib" ] || continue
    RSMP_AMD_LIB=$PWD/$lib python3 bench.py --no-cpu --no-secondary --steps 40 --warmup 5 "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-46s round $rep  ms/step %.4f  kernel %.4f  frac %.4f' % ('$lib', d['ms_per_step'], r['kernel_ms'], r['frac']))"
  done
done
