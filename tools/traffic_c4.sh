#!/bin/bash
# tools/traffic_c4.sh TAG -- GPU box: config 4's HBM traffic alone (separate FETCH_SIZE / WRITE_SIZE passes over bench.py --config c4, as
# tools/profile_round.sh runs them) and tools/profile_summary.py over them: gpurun_out/TAG/summary/{bench_c4.json, traffic_c4.json, pmc_c4.txt}.
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:?run through gpurun}
OUT="$R/gpurun_out/$TAG"; SUM="$OUT/summary"; mkdir -p "$SUM"
cd /tmp && export TMPDIR=/tmp
python3 "$R/bench.py" --config c4 --steps 16384 --warmup 512 > "$SUM/bench_c4.json" 2> "$OUT/bench_c4.err"
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 5 600 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_c4_$c" -- python3 "$R/bench.py" --config c4 --steps 512 --warmup 256 --spinup-seconds 0 > /dev/null 2> "$OUT/pmc_c4_$c.err"
done
python3 "$R/tools/profile_summary.py" "$OUT" "$SUM" > /dev/null
cat "$SUM/traffic_c4.json"
find "$OUT" -mindepth 1 -maxdepth 1 ! -name summary -exec rm -rf {} +
