#!/bin/bash
# tools/split_iter.sh [tag] -- one iteration on the split FIR kernel (GPU box, through gpurun): parity of the FIR paths, the
# headline bench line without CPU / secondary legs, and the per-wave phase clocks of the diagnostic build.
TAG=${1:-it}
O=gpurun_out/$TAG
mkdir -p $O
python -m pytest tests/test_fir_gpu.py -m gpu -x -q 2>&1 | tail -8 > $O/tests.log
python bench.py --no-cpu --no-secondary > $O/bench.json 2> $O/bench.err
RSMP_FIR_WTRACE=$O/wtrace.txt python bench.py --no-cpu --no-secondary --steps 3 --warmup 1 --spinup-seconds 0 > /dev/null 2>&1
python tools/wphase_report.py $O/wtrace.txt 111.5 > $O/wphase.txt 2>/dev/null
rm -f $O/wtrace.txt
tail -3 $O/tests.log
python -c "
import json; d=json.load(open('$O/bench.json')); print('ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])"
cat $O/wphase.txt
