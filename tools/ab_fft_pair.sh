#!/bin/bash
# tools/ab_fft_pair.sh [reps] -- GPU box: the FFT bench line with the two-channel kernel (fft_pair.hip) and, through the
# RSMP_FFT_PAIR=0 knob, with the wave-per-channel kernel (fft_wave.hip), interleaved `reps` times in one lease.
REPS=${1:-3}
for rep in $(seq $REPS); do
  for k in 1 0; do
    RSMP_DEBUG=1 RSMP_FFT_PAIR=$k timeout -k 5 120 python bench.py --path fft --no-cpu --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('pair=$k  ms/step %.4f kernel %.4f (median %.4f max %.4f) frac %.4f' % (d['ms_per_step'], r['kernel_ms'], r['kernel_ms_median'], r['kernel_ms_max'], r['frac']))"
  done
done
