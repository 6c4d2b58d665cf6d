#!/bin/bash
# tools/ab_channels.sh -- tools/channels_bench.py for the shipping library and every A/B library, same lease
for lib in resampler_amd/libresampler_amd.so resampler_amd/libresampler_amd_exp*.so; do
  echo "== $lib"
  RSMP_AMD_LIB=$PWD/$lib timeout -k 5 200 python tools/channels_bench.py 2>&1 | grep " ch "
done
