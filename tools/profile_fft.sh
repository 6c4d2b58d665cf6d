#!/bin/bash
# tools/profile_fft.sh <tag> -- the FFT part of tools/profile_round.sh alone (GPU box, through gpurun): bench line, kernel stats,
# HBM traffic and SQ counters of `bench.py --path fft`, the rate-pair and channel-count tables, the wave timeline.  Summaries in
# gpurun_out/<tag>/summary/ (copy to profiles/<tag>/; merge the "fft" entry of traffic_latest.json by hand).
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:?run through gpurun}
OUT="$R/gpurun_out/$TAG"
SUM="$OUT/summary"
mkdir -p "$SUM"
cd /tmp && export TMPDIR=/tmp
python3 "$R/bench.py" --path fft --steps 20 --warmup 3 > "$SUM/bench_fft.json" 2> "$OUT/bench_fft.err"
python3 "$R/tools/fft_channels_bench.py" > "$SUM/fft_channels_bench.txt" 2> "$OUT/fft_channels_bench.err"
python3 "$R/tools/fft_pairs_bench.py" --all > "$SUM/fft_pairs_bench.txt" 2> "$OUT/fft_pairs_bench.err"
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_fft" -- python3 "$R/bench.py" --path fft --steps 20 --warmup 3 --no-cpu \
    > "$SUM/bench_fft_under_rocprofv3.json" 2> "$OUT/kt_fft.err"
cp "$(find "$OUT/kt_fft" -name '*kernel_stats.csv' | head -1)" "$SUM/fft_kernel_stats.csv" 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 5 600 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_fft_$c" -- python3 "$R/bench.py" --path fft --steps 3 --warmup 1 --no-cpu > /dev/null 2> "$OUT/pmc_fft_$c.err"
done
for grp in "SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INST_LEVEL_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU" \
           "GRBM_GUI_ACTIVE"; do
    n=$(echo $grp | tr ' ' '_' | cut -c1-40)
    timeout -k 5 600 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_fft_$n" -- python3 "$R/bench.py" --path fft --steps 3 --warmup 1 --no-cpu > /dev/null 2> "$OUT/pmc_fft_$n.err"
done
python3 "$R/tools/profile_summary.py" "$OUT" "$SUM" > /dev/null 2>&1
rm -f "$SUM"/traffic_fir.json "$SUM"/traffic_c4.json "$SUM"/traffic_c5.json "$SUM"/pmc_fir.txt "$SUM"/pmc_c4.txt "$SUM"/pmc_c5.txt
find "$OUT" -mindepth 1 -maxdepth 1 ! -name summary -exec rm -rf {} +
