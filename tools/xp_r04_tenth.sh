#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_tenth
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_kdyn_gpu.py -m gpu -x -q > $O/pytest_kdyn.log 2>&1; tail -4 $O/pytest_kdyn.log
for t in 0 1; do
  SMO_KD_DENSE_TAIL=$t timeout -k 10 600 python bench.py --npts 256 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench256_tail$t.json 2> $O/bench256_tail$t.err || tail -3 $O/bench256_tail$t.err
  python3 -c "
import json; d=json.load(open('$O/bench256_tail$t.json')); c=d['config']; r=d['roofline']
print('tail=$t ms/step %.1f dev %.1f stack %.1f GB ck %d J %r frac %.4f sampled %.1f every %.1f' % (d['ms_per_step'], c['value_device_vectors']['ms_per_step'], c['stack_GB'], c['checkpoint_interval'], c['J'], r['frac'], 1e3*r['avg_launch_ms_sampled'], 1e3*r['avg_launch_ms_every_launch']))
print([ (k['kernel'], round(1e3*k['avg_ms'],1), k['launches']) for k in r['all_kernels']])"
done
