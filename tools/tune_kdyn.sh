#!/bin/bash
# sweep of the KDyn tile knobs (env SMO_KD_*): prints per-kernel average microseconds at 128^3
run() {
  env "$@" python bench.py --iters 60 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('$*', 'ms/steppair %.4f' % (d['ms_per_step']/60), ' '.join('%s=%.1f' % (k['kernel'].replace('kd_',''), k['avg_ms']*1e3) for k in d['roofline']['all_kernels'] if k['launches']>20))"
}
run SMO_KD_XT=16
run SMO_KD_XT=8
run SMO_KD_XT=8 SMO_KD_XTA=4
run SMO_KD_XT=8 SMO_KD_YT=8
run SMO_KD_XT=8 SMO_KD_ZT=2
