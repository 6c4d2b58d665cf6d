#!/bin/bash
# tools/fft_traffic.sh [lib ...] -- GPU box: HBM traffic (FETCH_SIZE x 2 on gfx950, WRITE_SIZE; separate --pmc passes) and kernel
# time of the FFT bench launch for the shipping library and the A/B builds named (default: every libresampler_amd_exp*.so).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run through gpurun}
LIBS="$R/resampler_amd/libresampler_amd.so ${@:-$(ls $R/resampler_amd/libresampler_amd_exp*.so 2>/dev/null)}"
for lib in $LIBS; do
  n=$(basename $lib .so)
  for c in FETCH_SIZE WRITE_SIZE; do
    RSMP_AMD_LIB=$lib rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/ffttr_${n}_$c -- python3 $R/bench.py --path fft --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
  done
  RSMP_AMD_LIB=$lib python3 - $R $n <<'PY'
import csv, glob, sys
root, n = sys.argv[1], sys.argv[2]
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for f in glob.glob(f"{root}/gpurun_out/ffttr_{n}_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "fft_ola" in r["Kernel_Name"] and r["Counter_Name"] == c:
                v.append(float(r["Counter_Value"]))
    tot[c] = sum(v) / max(1, len(v))
fetch, write = 2 * tot["FETCH_SIZE"] * 1024, tot["WRITE_SIZE"] * 1024
alg = 1121665024
print(f"{n:32s} fetch {fetch / 1e6:8.1f} MB  write {write / 1e6:8.1f} MB  total / algorithmic {(fetch + write) / alg:.3f}")
PY
done
