#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_11
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_poiseuille_gpu.py -m gpu -x -q > $O/pytest_pois.log 2>&1; tail -4 $O/pytest_pois.log
for f in 0 1 0 1; do
  SMO_POIS_XFUSE=$f timeout -k 10 300 python bench.py --workload pois --steps 3 --warmup 1 > $O/pois_fuse$f.json 2> $O/pois_fuse$f.err || tail -3 $O/pois_fuse$f.err
  python3 -c "
import json; d=json.load(open('$O/pois_fuse$f.json')); print('xfuse=$f value %.3f ms %.1f J %r match %s' % (d['value'], d['ms_per_step'], d['config']['J'], d['config'].get('J_matches_oracle_1e-6')), [(k['kernel'][:24], k['launches'], round(1e3*k['avg_ms'],1)) for k in d['roofline']['all_kernels']])"
done
timeout -k 10 900 python bench.py --npts 256 --iters 100 --steps 1 --warmup 1 --no-secondary > $O/bench_cpu256.json 2> $O/bench_cpu256.err; python3 -c "
import json; d=json.load(open('$O/bench_cpu256.json')); print(d['cpu_baseline']); print(d['config']['cpu_single_socket'])"
timeout -k 10 600 python bench.py --npts 128 --iters 100 --steps 1 --warmup 1 --no-secondary > $O/bench_cpu128.json 2> $O/bench_cpu128.err; python3 -c "
import json; d=json.load(open('$O/bench_cpu128.json')); print(d['cpu_baseline']); print(d['config']['cpu_single_socket'])"
