#!/bin/bash
# tools/knob_sweep.sh -- parity of the A/B switches that are left in the library (each is read once per process, and only
# under RSMP_DEBUG=1): the split FIR kernel's geometry switches, the FFT wave kernel's build switches, the lock-step
# kernel's exact-f32 switch.  One line per run; every "bad" must be 0 and every rms ~1e-7.  GPU box.  The same runs are
# tests/test_knobs_gpu.py.
export RSMP_DEBUG=1
run() { echo "== $1"; env $1 timeout -k 5 150 python tools/split_geo_check.py $2 2>&1 | grep -v amdgpu.ids | awk '{print "  ", $0}' | cut -c1-150; }
run RSMP_FIR_SPLIT_LONG=0 "2 96000 44100 2 48000 96000"
run RSMP_FIR_SPLIT_WIDE=0 "8 44100 48000 1 48000 44100"
run RSMP_FIR_SPLIT_QUADS=0 "8 96000 44100 16 96000 44100"
run RSMP_FIR_SPLIT_PLANES=3 "2 44100 48000 2 48000 44100"
run RSMP_FIR_MFMA=0 "2 44100 48000 4 48000 44100"
run RSMP_FIR_MFMA=1 "2 44100 48000 2 96000 44100"
run RSMP_FIR_MFMA_RING=1 "2 44100 48000"
fft() { echo "== $1"; env $1 timeout -k 5 150 python tools/fft_pair_check.py $2 2>&1 | grep -v amdgpu.ids | tail -4 | awk '{print "  ", $0}' | cut -c1-150; }
fft RSMP_FFT_WAVE_NOC2=1 "44100 48000 2 25 44100 48000 4 20"
fft RSMP_FFT_WAVE_WIDE=3 "22050 96000 2 12 88200 96000 2 12"
fft RSMP_FFT_WAVE=0 "44100 48000 2 25"
fft RSMP_FFT_GENERIC=1 "44100 48000 2 25"
# (two tests assert that workgroups DO run the split arithmetic by default: not under this knob)
echo "== RSMP_LS_EXACT=1"; RSMP_LS_EXACT=1 timeout -k 5 600 python -m pytest tests/test_fir_lockstep_gpu.py -q -m gpu -k "not mixed_batches and not runs_split" 2>&1 | tail -1
