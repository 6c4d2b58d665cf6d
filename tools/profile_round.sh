#!/bin/bash
# tools/profile_round.sh <tag> -- collects the rocprofv3 evidence kept under profiles/<tag>/ (run on the GPU
# box through gpurun; raw output goes to gpurun_out/<tag>/, the summaries to gpurun_out/<tag>/summary/, which
# is what gets copied to profiles/<tag>/):
#   1. bench.py as the driver runs it (headline line incl. cpu_baseline and `secondary`)     -> bench_n1.json
#      bench.py --path fft / --config c4 / --config c5                                        -> bench_fft.json, bench_c4.json, bench_c5.json
#      per-call latency of config 5, other channel counts, all 90 FFT rate pairs              -> bench_c5_calls.json, channels_bench.txt, fft_channels_bench.txt, fft_pairs_bench.txt
#      config 4's run: timelines (tools/kt_timeline.py), phase clocks, 16 steps per launch        -> c4_run_timeline_*.txt, wphase_c4_run_*.txt, bench_c4_k16.json
#   2. rocprofv3 --kernel-trace --stats of the three commands                                 -> *_kernel_stats.csv
#   3. separate --pmc passes FETCH_SIZE, WRITE_SIZE per workload (guide's gfx950 correction)  -> traffic_*.json, traffic_latest.json
#   4. separate --pmc passes: SQ / MFMA / LDS counters per workload                           -> pmc_*.txt
# The program after `--` is python3 itself (no env/bash hop); counters never share a run with a trace.
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repository copy on the GPU box)}
OUT="$R/gpurun_out/$TAG"
SUM="$OUT/summary"
mkdir -p "$SUM"
cd /tmp && export TMPDIR=/tmp

python3 "$R/bench.py" --steps 20 --warmup 3 > "$SUM/bench_n1.json" 2> "$OUT/bench_n1.err"
python3 "$R/bench.py" --path fft --steps 20 --warmup 3 > "$SUM/bench_fft.json" 2> "$OUT/bench_fft.err"
python3 "$R/bench.py" --config c4 --steps 16384 --warmup 512 > "$SUM/bench_c4.json" 2> "$OUT/bench_c4.err"
python3 "$R/bench.py" --config c4 --c4-k 1 --steps 256 --warmup 16 > "$SUM/bench_c4_k1.json" 2> "$OUT/bench_c4_k1.err"
python3 "$R/bench.py" --config c4 --c4-k 16 --steps 2048 --warmup 256 > "$SUM/bench_c4_k16.json" 2> "$OUT/bench_c4_k16.err"
python3 "$R/tools/distinct_probe.py" 64 > "$SUM/distinct_states_probe.txt" 2> "$OUT/distinct_probe.err"
(cd "$R" && tools/c4_shard_sweep.sh > "$SUM/c4_shard_sweep.txt" 2> "$OUT/c4_shard_sweep.err")
python3 "$R/bench.py" --config c5 --steps 10 --warmup 2 > "$SUM/bench_c5.json" 2> "$OUT/bench_c5.err"
# the RCCL exchange on what hardware there is: a world of one rank sending to / receiving from itself
timeout -k 5 300 python3 "$R/bench.py" --config c4 --feed rccl --gpus 1 --steps 256 --warmup 16 > "$SUM/bench_c4_rccl_world1.json" 2> "$OUT/bench_c4_rccl.err"
python3 "$R/tools/configs_bench.py" > "$SUM/bench_c5_calls.json" 2> "$OUT/bench_c5_calls.err"
python3 "$R/tools/channels_bench.py" > "$SUM/channels_bench.txt" 2> "$OUT/channels_bench.err"
python3 "$R/tools/fft_channels_bench.py" > "$SUM/fft_channels_bench.txt" 2> "$OUT/fft_channels_bench.err"
python3 "$R/tools/fft_pairs_bench.py" --all > "$SUM/fft_pairs_bench.txt" 2> "$OUT/fft_pairs_bench.err"
RSMP_DEBUG=1 RSMP_LS_TRACE="$SUM/ls_trace_raw.txt" python3 "$R/tools/ls_trace.py" > "$SUM/ls_trace.txt" 2> "$OUT/ls_trace.err"; rm -f "$SUM/ls_trace_raw.txt"
RSMP_DEBUG=1 RSMP_FIR_WTRACE="$OUT/wtrace.txt" python3 "$R/bench.py" --no-cpu --no-secondary --steps 3 --warmup 1 --spinup-seconds 0 > /dev/null 2>&1
python3 "$R/tools/wphase_report.py" "$OUT/wtrace.txt" 111.5 > "$SUM/wphase_split.txt" 2>/dev/null
(cd "$R" && tools/c5_trace.sh > /dev/null 2>&1; cp gpurun_out/c5trace/wphase_*.txt "$SUM/" 2>/dev/null)
# config 4's run: queue / start / duration of the last kernels of a run of 256 and of 16 calls (planned ahead: q of the plan
# stream next to the caller's), the same without plan-ahead, and the phase clocks of the two-round kernel inside a run
for k in 256 16; do
    timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt_run$k" -- python3 "$R/tools/run_probe.py" 1024 $k > /dev/null 2> "$OUT/kt_run$k.err"
    python3 "$R/tools/kt_timeline.py" "$(find "$OUT/kt_run$k" -name '*kernel_trace.csv' | head -1)" 30 > "$SUM/c4_run_timeline_k$k.txt" 2>/dev/null
done
RSMP_DEBUG=1 RSMP_LS_AHEAD=0 timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt_run_noahead" -- python3 "$R/tools/run_probe.py" 1024 256 > /dev/null 2> "$OUT/kt_run_noahead.err"
python3 "$R/tools/kt_timeline.py" "$(find "$OUT/kt_run_noahead" -name '*kernel_trace.csv' | head -1)" 24 > "$SUM/c4_run_timeline_k256_no_plan_ahead.txt" 2>/dev/null
for k in 256 16; do
    RSMP_DEBUG=1 RSMP_LS_AHEAD=0 RSMP_FIR_WTRACE="$OUT/w_c4.raw" python3 "$R/tools/run_probe.py" 1024 $k > /dev/null 2>&1
    python3 "$R/tools/wphase_report.py" "$OUT/w_c4.raw" $([ $k = 256 ] && echo 36 || echo 2.25) > "$SUM/wphase_c4_run_k$k.txt" 2>/dev/null
done

(cd "$R" && tools/c4_timelines.sh "$TAG/tl" > /dev/null 2>&1; cp "$OUT/tl/"c4_run_timeline_*_ahead*.txt "$SUM/" 2>/dev/null)
timeout -k 5 900 python3 "$R/tools/soak_lockstep.py" --hours 24 > "$SUM/soak_lockstep_24h.txt" 2> "$OUT/soak24.err"

declare -A CMD
CMD[fir]="--steps 20 --warmup 3 --no-cpu --no-secondary"
CMD[fft]="--path fft --steps 20 --warmup 3 --no-cpu"
CMD[c4]="--config c4 --steps 16384 --warmup 512"
CMD[c5]="--config c5 --steps 10 --warmup 2"
for w in fir fft c4 c5; do
    timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$w" -- python3 "$R/bench.py" ${CMD[$w]} \
        > "$SUM/bench_${w}_under_rocprofv3.json" 2> "$OUT/kt_$w.err"
    cp "$(find "$OUT/kt_$w" -name '*kernel_stats.csv' | head -1)" "$SUM/${w}_kernel_stats.csv" 2>/dev/null
done

declare -A PCMD
PCMD[fir]="--steps 3 --warmup 1 --no-cpu --no-secondary"
PCMD[fft]="--path fft --steps 3 --warmup 1 --no-cpu"
PCMD[c4]="--config c4 --steps 512 --warmup 256 --spinup-seconds 0"
PCMD[c5]="--config c5 --steps 2 --warmup 1 --spinup-seconds 0"
for w in fir fft c4 c5; do
    for c in FETCH_SIZE WRITE_SIZE; do
        timeout -k 5 600 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_${w}_$c" -- python3 "$R/bench.py" ${PCMD[$w]} \
            > /dev/null 2> "$OUT/pmc_${w}_$c.err"
    done
    for grp in "SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
               "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
               "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INST_LEVEL_LDS" \
               "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES" \
               "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU" \
               "GRBM_GUI_ACTIVE"; do
        n=$(echo $grp | tr ' ' '_' | cut -c1-40)
        [ "$w" = c5 ] && continue   # (config 5: traffic counters only)
        timeout -k 5 600 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_${w}_$n" -- python3 "$R/bench.py" ${PCMD[$w]} \
            > /dev/null 2> "$OUT/pmc_${w}_$n.err"
    done
done
python3 "$R/tools/profile_summary.py" "$OUT" "$SUM"
# only the summaries travel back (gpurun merges at most 64 MiB)
find "$OUT" -mindepth 1 -maxdepth 1 ! -name summary -exec rm -rf {} +
