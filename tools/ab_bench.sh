#!/bin/bash
# tools/ab_bench.sh <bench args...> -- the same bench line for every A/B library resampler_amd/libresampler_amd_exp*.so
# (make -C resampler_amd/csrc exp EXPS=...) and the shipping one; prints ms per step / kernel ms / roofline fraction.
for lib in resampler_amd/libresampler_amd.so resampler_amd/libresampler_amd_exp*.so; do
  for rep in 1 2; do
    RSMP_AMD_LIB=$PWD/$lib python bench.py --no-cpu --no-secondary "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-52s ms/step %.4f kernel %.4f frac %.4f' % ('$lib', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
  done
done
