#!/bin/bash
# tools/ab_headline.sh -- VERDICT r03 item 2(a): the headline bench line for three libraries inside ONE lease: round 2's
# HEAD (dcf032a -> resampler_amd/libresampler_amd_expr02.so), the block-floating-point commit (36451ca -> _expbfp.so) and
# this tree's; built with `git worktree add build/wt_X <commit> && make -C build/wt_X/resampler_amd/csrc ../libresampler_amd.so`.
# Three rounds, interleaved, so that a drifting box shows up as drift and not as a difference between the libraries.
for rep in 1 2 3; do
  for lib in resampler_amd/libresampler_amd_expr02.so resampler_amd/libresampler_amd_expbfp.so resampler_amd/libresampler_amd.so; do
    [ -f "$lib" ] || continue
    RSMP_AMD_LIB=$PWD/$lib python3 bench.py --no-cpu --no-secondary --steps 40 --warmup 5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-46s round $rep  ms/step %.4f  kernel %.4f  frac %.4f  step-kernel %.1f us' % ('$lib', d['ms_per_step'], r['kernel_ms'], r['frac'], (d['ms_per_step'] - r['kernel_ms']) * 1e3))"
  done
done
