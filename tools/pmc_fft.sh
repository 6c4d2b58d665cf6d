# usage (GPU box): bash tools/pmc_fft.sh -- SQ / LDS counters of fft_ola_kernel for bench.py --path fft
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run through gpurun}
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/fftpmc_$n -- python3 $R/bench.py --path fft --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
done
python3 - <<'PY'
import csv,glob,os,collections
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out'
tot=collections.defaultdict(list)
for f in glob.glob(root+'/fftpmc_*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'fft_ola' in r['Kernel_Name']:
            tot[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(tot.items()):
    print(f"{k:28s} {sum(v)/len(v):18.1f}  n={len(v)}")
PY
