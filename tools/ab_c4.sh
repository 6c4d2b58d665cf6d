#!/bin/bash
# tools/ab_c4.sh -- config 4 (lock-step) step time for the shipping library and every A/B library, same lease
for lib in resampler_amd/libresampler_amd.so resampler_amd/libresampler_amd_exp*.so; do
  for rep in 1 2; do
    RSMP_AMD_LIB=$PWD/$lib python bench.py --config c4 --steps 256 --warmup 16 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-52s ms/step %.5f kernel %.5f wgs %s' % ('$lib', d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['workgroups_this_rank']))"
  done
done
