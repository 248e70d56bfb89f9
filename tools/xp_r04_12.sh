#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_12
mkdir -p $O
cd $R
for cfg in "2,256" "1,256" "1,512" "1,1024" "2,512" "2,1024"; do
  SMO_POIS_XFUSE_CFG=$cfg timeout -k 10 300 python bench.py --workload pois --steps 3 --warmup 1 > $O/pois_$cfg.json 2> $O/pois_$cfg.err || tail -3 $O/pois_$cfg.err
  python3 -c "
import json; d=json.load(open('$O/pois_$cfg.json')); print('cfg=$cfg value %.3f ms %.1f match %s' % (d['value'], d['ms_per_step'], d['config'].get('J_matches_oracle_1e-6')), [(k['kernel'][:12], k['launches'], round(1e3*k['avg_ms'],1)) for k in d['roofline']['all_kernels'] if k['launches']>100])"
done
