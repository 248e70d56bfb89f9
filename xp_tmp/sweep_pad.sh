#!/bin/bash
# usage: sweep_pad.sh "<pads>" [npts] [iters]
N=${2:-128}; IT=${3:-200}
for p in $1; do
  SMO_KD_TYPAD=$p timeout -k 10 200 python bench.py --npts $N --iters $IT --steps 2 --warmup 1 --no-secondary --no-cpu-baseline > gpurun_out/pad_$p.json 2> gpurun_out/pad_$p.err || { echo "pad $p failed"; tail -3 gpurun_out/pad_$p.err; exit 1; }
  python - <<PY
import json
d=json.loads(open('gpurun_out/pad_$p.json').read().strip().splitlines()[-1])
print('pad $p: %.2f ms/step J=%r'%(d['ms_per_step'], d['config']['J']), ' '.join('%s=%.1f'%(k['kernel'].replace('kd_','').replace('_pass',''),k['avg_ms']*1e3) for k in d['roofline']['all_kernels'][:8]))
PY
done
