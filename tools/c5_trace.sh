#!/bin/bash
# tools/c5_trace.sh -- per-wave phase clocks of the two-round split kernel on config 5's stream (8 ch) and on its
# two-channel sibling (diagnostic instantiation, RSMP_FIR_WTRACE), one short bulk launch each.
O=gpurun_out/c5trace; mkdir -p $O
RSMP_DEBUG=1 RSMP_FIR_WTRACE=$PWD/$O/w8.txt timeout -k 5 100 python bench.py --config c5 --c5-frames 5760000 --steps 2 --warmup 1 --spinup-seconds 0 > /dev/null 2>$O/err8.txt
python tools/wphase_report.py $O/w8.txt 17.6 > $O/wphase_c5_8ch.txt; cat $O/wphase_c5_8ch.txt
RSMP_DEBUG=1 RSMP_FIR_WTRACE=$PWD/$O/w2.txt timeout -k 5 100 python - <<'PY' 2>$O/err2.txt
import torch, resampler_amd as ra
from resampler_amd import synth
dev = torch.device("cuda:0")
hs = [ra.ResamplerFir.new_from_hz(2, 96000, 44100, ra.Latency.Sample64, ra.Attenuation.Db120) for _ in range(64)]
x = torch.from_numpy(synth.fast_noise(2 << 20, seed=2)).to(dev)
d_in = [(x * (0.5 + 0.5 * i / 64)).contiguous() for i in range(64)]
d_out = [torch.empty(hs[0].bulk_output_bound(2 << 20, 1024), device=dev) for _ in hs]
b = ra.FirBatch(hs); b.bind(d_in, d_out)
for _ in range(3):
    b.reset(); b.resample_bulk_device(1024, ra.torch_stream())
torch.cuda.synchronize()
PY
python tools/wphase_report.py $O/w2.txt 51.2 > $O/wphase_2ch_96_441.txt; cat $O/wphase_2ch_96_441.txt
rm -f $O/w8.txt $O/w2.txt
