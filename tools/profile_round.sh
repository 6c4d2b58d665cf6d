#!/bin/bash
# tools/profile_round.sh <tag> -- collects the rocprofv3 evidence kept under profiles/<tag>/ for the
# FIR headline bench (run on the GPU box through gpurun; raw output goes to gpurun_out/<tag>/):
#   1. bench.py as is                                   -> bench_n1.json
#   2. rocprofv3 --kernel-trace --stats of the same run -> bench_kernel_stats.csv, bench_under_rocprofv3.json
#   3. separate --pmc passes: FETCH_SIZE, WRITE_SIZE    -> traffic_fir.json  (guide's gfx950 correction)
#   4. separate --pmc passes: SQ / MFMA / LDS counters  -> pmc_fir.txt
# The program after `--` is python3 itself (no env/bash hop), counters never share a run with a trace.
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp

python3 "$R/bench.py" --steps 20 --warmup 3 > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"

rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu \
    > "$OUT/bench_under_rocprofv3.json" 2> "$OUT/kt.err"
cp "$(find "$OUT/kt" -name '*kernel_stats.csv' | head -1)" "$OUT/bench_kernel_stats.csv" 2>/dev/null

for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu \
        > /dev/null 2> "$OUT/pmc_$c.err"
done
for grp in "SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES" \
           "GRBM_GUI_ACTIVE"; do
    n=$(echo $grp | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$n" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu \
        > /dev/null 2> "$OUT/pmc_$n.err"
done

python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(list)
for f in glob.glob(out + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'fir_periodic' in r['Kernel_Name'] or 'fir_split' in r['Kernel_Name']:
            tot[r['Counter_Name']].append(float(r['Counter_Value']))
mean = {k: sum(v) / len(v) for k, v in tot.items()}
bench = json.loads(open(out + '/bench_n1.json').read().strip().splitlines()[-1])
alg = bench['roofline']['algorithmic_bytes']
if 'FETCH_SIZE' in mean and 'WRITE_SIZE' in mean:
    fetch = mean['FETCH_SIZE'] * 1024 * 2      # gfx950: FETCH_SIZE tallies 128-B requests at 64 B
    write = mean['WRITE_SIZE'] * 1024
    json.dump({
        'kernel': bench['roofline']['kernel'],
        'command': 'python3 bench.py --steps 3 --warmup 1 --no-cpu (one rocprofv3 --pmc pass per counter)',
        'raw': {'FETCH_SIZE': {'dispatches': len(tot['FETCH_SIZE']), 'mean_kb': mean['FETCH_SIZE']},
                'WRITE_SIZE': {'dispatches': len(tot['WRITE_SIZE']), 'mean_kb': mean['WRITE_SIZE']}},
        'correction': 'FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md: a coalesced stream is tallied at 1/2); WRITE_SIZE as is',
        'fetch_bytes_per_launch': fetch, 'write_bytes_per_launch': write,
        'hbm_bytes_per_launch': fetch + write, 'algorithmic_bytes_per_launch': alg,
        'ratio_traffic_to_algorithmic': (fetch + write) / alg}, open(out + '/traffic_fir.json', 'w'), indent=1)
with open(out + '/pmc_fir.txt', 'w') as f:
    f.write('%s, bench.py --steps 3 --warmup 1 --no-cpu, per-dispatch means (rocprofv3 --pmc, separate passes)\n'
            % bench['roofline']['kernel'])
    for k in sorted(mean):
        f.write('%-28s %18.1f   n=%d\n' % (k, mean[k], len(tot[k])))
    if 'GRBM_GUI_ACTIVE' in mean:
        cyc = mean['GRBM_GUI_ACTIVE'] / 8
        f.write('cycles per dispatch (GRBM_GUI_ACTIVE / 8 XCDs): %.0f\n' % cyc)
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in mean:
            f.write('matrix pipe busy: %.1f %% of 1024 SIMDs x cycles\n' % (100 * mean['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc)))
print(open(out + '/pmc_fir.txt').read())
print(open(out + '/bench_n1.json').read())
PY
