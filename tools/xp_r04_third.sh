#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_third
mkdir -p $O
cd $R
for st in 1 0; do for stride in 8 1; do
  SMO_TIMING_STAMP=$st SMO_BENCH_TIMING_STRIDE=$stride timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench_stamp${st}_s$stride.json 2> $O/bench_stamp${st}_s$stride.err || { echo "bench failed"; tail -5 $O/bench_stamp${st}_s$stride.err; }
  python3 - <<PY
import json
d=json.load(open("$O/bench_stamp${st}_s$stride.json")); r=d["roofline"]
print("stamp=$st stride=$stride", "ms/step %.2f"%d["ms_per_step"], "dev %.2f"%d["config"]["value_device_vectors"]["ms_per_step"], "sampled %.2f"%(1e3*r["avg_launch_ms_sampled"]), "every %.2f"%(1e3*r["avg_launch_ms_every_launch"]), "frac %.4f"%r["frac"], "sum_every %.1f"%(1e3*sum(k["avg_ms"]*k["launches"] for k in r["all_kernels"])/1000))
PY
done; done
echo "== z swizzles 128"; bash tools/sweep_variants.sh "base zswz1 zswz2 base" 128 200
echo "== z swizzles 256"; bash tools/sweep_variants.sh "base zswz1 zswz2" 256 16
timeout -k 10 900 python bench.py --devices 0,0,0,0,0,0,0,0 --npts 256 --iters 50 --steps 1 --warmup 1 > $O/bench_dev8_256.json 2> $O/bench_dev8_256.err; python3 -c "
import json; d=json.load(open('$O/bench_dev8_256.json')); c=d['config']; print({k:c[k] for k in ('compute_ms_per_step_pair','exchange_ms_per_step_pair','wall_ms_per_step_pair','host_issue_ms_per_step_pair','host_bound_loop','transpose_pull')})"
timeout -k 10 900 python bench.py --devices 0,0 --npts 256 --iters 50 --steps 1 --warmup 1 > $O/bench_dev2_256.json 2> $O/bench_dev2_256.err; python3 -c "
import json; d=json.load(open('$O/bench_dev2_256.json')); c=d['config']; print({k:c[k] for k in ('compute_ms_per_step_pair','exchange_ms_per_step_pair','wall_ms_per_step_pair','host_issue_ms_per_step_pair','host_bound_loop')})"
timeout -k 10 900 python -m pytest tests/test_bench_gpu.py tests/test_kdyn_multi_device_gpu.py -m gpu -x -q -k "single_process or kdyn_line or reference_callbacks or distributed_device" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
