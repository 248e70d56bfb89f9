#!/bin/bash
# Ty layout A/B on the real kernels: plane layout (SMO_KD_TYL=0) against z-block major with the y pass's workgroups kx-fastest / z-fastest.
#   tools/sweep_ty_layout.sh [iters]      -> gpurun_out/ty_layout_sweep.txt (per-kernel microseconds from bench.py's warm-up breakdown)
IT=${1:-200}
mkdir -p gpurun_out
out=gpurun_out/ty_layout_sweep.txt; : > $out
for N in 128 256; do
  for cfg in "0 1" "1 1" "1 0"; do
    set -- $cfg
    SMO_KD_TYL=$1 SMO_KD_YKX=$2 timeout -k 10 400 python bench.py --npts $N --iters $IT --steps 2 --warmup 1 --no-secondary --no-cpu-baseline --no-host-vectors > gpurun_out/tyl.json 2> gpurun_out/tyl.err || { echo "N=$N TYL=$1 failed" | tee -a $out; tail -3 gpurun_out/tyl.err | tee -a $out; exit 1; }
    python - "$N" "$1" "$2" <<'PY' | tee -a $out
import json, sys
d = json.loads(open('gpurun_out/tyl.json').read().strip().splitlines()[-1])
print('N=%s TYL=%s YKX=%s: %.2f ms/gradient J=%r | ' % (sys.argv[1], sys.argv[2], sys.argv[3], d['ms_per_step'], d['config']['J']) +
      ' '.join('%s=%.1f' % (k['kernel'].replace('kd_', '').replace('_pass', '').replace('_forward', 'f'), k['avg_ms'] * 1e3) for k in d['roofline']['all_kernels'][2:8]))
PY
  done
done
