#!/bin/bash
# Per-kernel timing of library variants on the GPU box (tuning aid; see DESIGN.md section 5 for what was measured with it).
#   tools/sweep_variants.sh "<names>" [npts] [iters]
# "base" = the in-tree libsmo.so; any other name N = xp_tmp/lib/libsmo_N.so (an experimental build, loaded through SMO_LIB;
# xp_tmp/ is git-ignored scratch that still travels with gpurun).  Environment knobs (e.g. SMO_KD_TYPAD) pass through.
N=${2:-128}; IT=${3:-200}
mkdir -p gpurun_out
for v in $1; do
  if [ "$v" = base ]; then unset SMO_LIB; else export SMO_LIB=$PWD/xp_tmp/lib/libsmo_$v.so; fi
  timeout -k 10 300 python bench.py --npts $N --iters $IT --steps 2 --warmup 1 --no-secondary --no-cpu-baseline --no-host-vectors > gpurun_out/v_$v.json 2> gpurun_out/v_$v.err || { echo "variant $v failed"; tail -3 gpurun_out/v_$v.err; exit 1; }
  python - <<PY
import json
d=json.loads(open('gpurun_out/v_$v.json').read().strip().splitlines()[-1])
print('$v: %.2f ms/step J=%r'%(d['ms_per_step'], d['config']['J']), ' '.join('%s=%.1f'%(k['kernel'].replace('kd_','').replace('_pass',''),k['avg_ms']*1e3) for k in d['roofline']['all_kernels'][:8]))
PY
done
