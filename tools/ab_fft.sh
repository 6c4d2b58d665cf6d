#!/bin/bash
# tools/ab_fft.sh [reps] -- GPU box: the FFT bench line (kernel ms by events) for the shipping library and every A/B build
# resampler_amd/libresampler_amd_exp*.so of fft_wave.hip (make -C resampler_amd/csrc exp EXPFILE=fft_wave.hip EXPS=...),
# interleaved `reps` times so that a drifting box shows; then the C3 parity test with each A/B build.
REPS=${1:-3}
for rep in $(seq $REPS); do
  for lib in resampler_amd/libresampler_amd.so resampler_amd/libresampler_amd_exp*.so; do
    RSMP_AMD_LIB=$PWD/$lib timeout -k 5 120 python bench.py --path fft --no-cpu --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-48s ms/step %.4f kernel %.4f (median %.4f max %.4f) frac %.4f' % ('$lib', d['ms_per_step'], r['kernel_ms'], r['kernel_ms_median'], r['kernel_ms_max'], r['frac']))"
  done
done
for lib in resampler_amd/libresampler_amd_exp*.so; do
  echo "parity $lib: $(RSMP_AMD_LIB=$PWD/$lib timeout -k 5 300 python -m pytest tests/test_fft_gpu.py -m gpu -x -q -k 'c3_full_size or bulk_equals_consecutive' 2>&1 | tail -1)"
done
