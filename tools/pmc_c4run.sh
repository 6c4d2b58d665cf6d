#!/bin/bash
# tools/pmc_c4run.sh [k] -- SQ counters of the run planner's chain kernel (bench.py --config c4 --c4-k k), one --pmc group per pass
set -u
K=${1:-256}
R=${GRAFT_REPO_ROOT:?run through gpurun}
O="$R/gpurun_out/pmc_c4run"
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 5 300 rocprofv3 --pmc $grp --output-format csv -d "$O/$n" -- python3 "$R/bench.py" --config c4 --c4-k $K --steps $((2*K)) --warmup 0 --spinup-seconds 0 > /dev/null 2> "$O/$n.err"
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"fir_lockstep_\w+", r["Kernel_Name"])
        if not m: continue
        k = m.group(0)
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k)
    for c, v in sorted(acc[k].items()):
        print("   %-22s %14.0f per launch" % (c, v / max(1, cnt[(k, c)])))
PY
