#!/bin/bash
# The grid stage in two staggered halves on two streams (SMO_KD_SPLIT=2) against the plain sequence: per-gradient time and J at 128^3 / 256^3.
#   tools/sweep_split.sh [iters]   -> gpurun_out/split_sweep.txt
IT=${1:-200}
mkdir -p gpurun_out
out=gpurun_out/split_sweep.txt; : > $out
for N in 128 256; do
  for S in 1 2 1 2; do
    SMO_KD_SPLIT=$S timeout -k 10 400 python bench.py --npts $N --iters $IT --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-host-vectors > gpurun_out/spl.json 2> gpurun_out/spl.err || { echo "N=$N SPLIT=$S failed" | tee -a $out; tail -3 gpurun_out/spl.err | tee -a $out; exit 1; }
    python - "$N" "$S" <<'PY' | tee -a $out
import json, sys
d = json.loads(open('gpurun_out/spl.json').read().strip().splitlines()[-1])
print('N=%s SPLIT=%s: %.2f ms/gradient J=%r' % (sys.argv[1], sys.argv[2], d['ms_per_step'], d['config']['J']))
PY
  done
done
