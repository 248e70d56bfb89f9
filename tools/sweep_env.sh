#!/bin/bash
# Per-kernel timing of an ENVIRONMENT KNOB (one library build) on the GPU box.  usage: tools/sweep_env.sh VAR "v1 v2 ..." [npts] [iters]
VAR=$1; N=${3:-128}; IT=${4:-200}
for v in $2; do
  export $VAR=$v
  timeout -k 10 300 python bench.py --npts $N --iters $IT --steps 2 --warmup 1 --no-secondary --no-cpu-baseline --no-host-vectors > gpurun_out/e_$v.json 2> gpurun_out/e_$v.err || { echo "$VAR=$v failed"; tail -3 gpurun_out/e_$v.err; exit 1; }
  python - <<PY
import json
d=json.loads(open('gpurun_out/e_$v.json').read().strip().splitlines()[-1])
print('$VAR=$v N=$N: %.2f ms/step J=%r'%(d['ms_per_step'], d['config']['J']), ' '.join('%s=%.1f'%(k['kernel'].replace('kd_','').replace('_pass',''),k['avg_ms']*1e3) for k in d['roofline']['all_kernels'][:8]))
PY
done
