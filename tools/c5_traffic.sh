#!/bin/bash
# tools/c5_traffic.sh -- HBM traffic of config 5's bulk launch: separate FETCH_SIZE / WRITE_SIZE passes (the guide's gfx950
# correction: FETCH_SIZE x 2) against the algorithmic bytes.  Run from the repository root on the GPU box.
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/c5traffic
cd /tmp && export TMPDIR=/tmp
rm -rf "$O"; mkdir -p "$O"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 rocprofv3 --pmc $c --output-format csv -d $O/$c -- python3 $R/bench.py --config c5 --steps 2 --warmup 1 --spinup-seconds 0 > $O/$c.json 2> $O/$c.err
done
python3 - <<PY
import csv, glob, json
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = [float(r["Counter_Value"]) for f in glob.glob("$O/%s/**/*counter_collection.csv" % c, recursive=True) for r in csv.DictReader(open(f)) if "fir_split" in r["Kernel_Name"] or "fir_periodic" in r["Kernel_Name"]]
    tot[c] = sum(v) / max(1, len(v))
b = json.loads(open("$O/FETCH_SIZE.json").read().strip().splitlines()[-1])
alg = b["roofline"]["algorithmic_bytes"]
hbm = tot["FETCH_SIZE"] * 1024 * 2 + tot["WRITE_SIZE"] * 1024
print("c5 traffic: fetch %.3f GB write %.3f GB total %.3f GB = %.2fx algorithmic (%.3f GB); kernel %s ms" % (tot["FETCH_SIZE"] * 2048 / 1e9, tot["WRITE_SIZE"] * 1024 / 1e9, hbm / 1e9, hbm / alg, alg / 1e9, b["roofline"]["kernel_ms"]))
PY
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
