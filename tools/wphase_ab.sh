#!/bin/bash
# tools/wphase_ab.sh -- the headline launch's per-wave phase clocks (RSMP_FIR_WTRACE) for the shipping library and every
# A/B library resampler_amd/libresampler_amd_exp*.so, one after the other on the same box.
O=gpurun_out/wphase_ab; mkdir -p $O
for lib in resampler_amd/libresampler_amd.so resampler_amd/libresampler_amd_exp*.so; do
  n=$(basename $lib .so)
  RSMP_DEBUG=1 RSMP_AMD_LIB=$PWD/$lib RSMP_FIR_WTRACE=$PWD/$O/$n.raw timeout -k 5 150 python bench.py --no-cpu --no-secondary --steps 3 --warmup 1 --spinup-seconds 0 > /dev/null 2>&1
  echo "== $n"; python tools/wphase_report.py $O/$n.raw 111.5 | tee $O/$n.txt; rm -f $O/$n.raw
done
