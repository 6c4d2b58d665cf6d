# usage: bash tools/clock_fir.sh  -- GPU clock held during the FIR kernel: GRBM_GUI_ACTIVE / 8 XCDs / kernel time.
# The counter and the durations come from SEPARATE runs (a --pmc pass serialises dispatches and must not share
# a run with a trace): pass 1 --kernel-trace, pass 2 --pmc GRBM_GUI_ACTIVE, same command.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run through gpurun}
for d in 0 16; do
  export RSMP_DEBUG=1 RSMP_FIR_DEBUG=$d
  rm -rf "$R/gpurun_out/clk_kt_$d" "$R/gpurun_out/clk_pmc_$d"
  rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/clk_kt_$d" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu --no-secondary > /dev/null 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$R/gpurun_out/clk_pmc_$d" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu --no-secondary > /dev/null 2>&1
  python3 - "$R/gpurun_out/clk_kt_$d" "$R/gpurun_out/clk_pmc_$d" "$d" <<'PY'
import csv, glob, sys
kt, pmc, d = sys.argv[1:4]
hit = lambda r: "fir_periodic" in r["Kernel_Name"] or "fir_split" in r["Kernel_Name"]
act = [float(r["Counter_Value"]) for f in glob.glob(pmc + "/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if hit(r)]
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for f in glob.glob(kt + "/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f)) if hit(r)]
a = sum(act) / len(act) / 8
t = sum(dur) / len(dur)
print("debug=%s  cycles/XCD %.0f  kernel %.1f us (unprofiled-counter run)  clock %.2f GHz" % (d, a, t / 1e3, a / t))
PY
done
