#!/bin/bash
# tools/split_iter.sh [tag] [pytest -k expr] -- one iteration on the split FIR kernel (GPU box, through gpurun): parity of the FIR
# paths, other rate pairs / channel counts (tools/channels_bench.py), config 5 and the headline line without CPU / secondary
# legs.  Every step under its own `timeout`: a kernel that hangs must not eat the lease.
TAG=${1:-it}
K=${2:-}
O=gpurun_out/$TAG
mkdir -p $O
if [ -n "$K" ]; then timeout -k 5 400 python -m pytest tests/test_fir_gpu.py -m gpu -x -q -k "$K" > $O/tests.log 2>&1
else timeout -k 5 400 python -m pytest tests/test_fir_gpu.py -m gpu -x -q > $O/tests.log 2>&1; fi
tail -12 $O/tests.log
timeout -k 5 200 python tools/channels_bench.py > $O/channels.txt 2>&1; cat $O/channels.txt
timeout -k 5 120 python bench.py --config c5 --steps 10 --warmup 2 > $O/bench_c5.json 2> $O/bench_c5.err
python -c "
import json; d=json.load(open('$O/bench_c5.json')); print('c5 ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel'], d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'lat', d['config']['chunk_latency_us'])"
timeout -k 5 120 python bench.py --no-cpu --no-secondary > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print('ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])"
