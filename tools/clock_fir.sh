# usage: bash tools/clock_fir.sh  -- GPU clock held during the FIR kernel: GRBM_GUI_ACTIVE / kernel time
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in 0 16; do
  export RSMP_FIR_DEBUG=$d
  rm -rf $R/gpurun_out/clk_$d
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/clk_$d -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu > /dev/null 2>&1
  python3 - <<PY
import csv,glob
root="$R/gpurun_out/clk_$d"
act=[float(r["Counter_Value"]) for f in glob.glob(root+"/**/*counter_collection.csv",recursive=True) for r in csv.DictReader(open(f)) if ("fir_periodic" in r["Kernel_Name"] or "fir_split" in r["Kernel_Name"])]
dur=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"])) for f in glob.glob(root+"/**/*kernel_trace.csv",recursive=True) for r in csv.DictReader(open(f)) if ("fir_periodic" in r["Kernel_Name"] or "fir_split" in r["Kernel_Name"])]
a=sum(act)/len(act)/8; t=sum(dur)/len(dur)
print("debug=$d  cycles/XCD %.0f  kernel %.1f us  clock %.2f GHz"%(a,t/1e3,a/t))
PY
done
