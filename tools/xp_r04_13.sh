#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_13
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_poiseuille_gpu.py -m gpu -x -q -k "fused or fixture or vs_oracle" > $O/pytest_pois.log 2>&1; tail -4 $O/pytest_pois.log
for f in 0 1 0 1; do
  SMO_POIS_XPROD=$f timeout -k 10 300 python bench.py --workload pois --steps 3 --warmup 1 > $O/pois_prod$f.json 2> $O/pois_prod$f.err || tail -3 $O/pois_prod$f.err
  python3 -c "
import json; d=json.load(open('$O/pois_prod$f.json')); print('xprod=$f value %.3f ms %.1f J %r match %s' % (d['value'], d['ms_per_step'], d['config']['J'], d['config'].get('J_matches_oracle_1e-6')), [(k['kernel'][:14], k['launches'], round(1e3*k['avg_ms'],1)) for k in d['roofline']['all_kernels'] if k['launches']>100])"
done
