#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_second
mkdir -p $O
cd $R
for pm in 1 0; do
  SMO_TIMING_PRE_MARKER=$pm timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench_pre$pm.json 2> $O/bench_pre$pm.err || { echo "bench pre$pm failed"; tail -5 $O/bench_pre$pm.err; }
  python3 - <<PY
import json
d=json.load(open("$O/bench_pre$pm.json")); r=d["roofline"]
print("pre_marker=$pm", "ms/step", d["ms_per_step"], "sampled", r["avg_launch_ms_sampled"], "every", r["avg_launch_ms_every_launch"], "frac", r["frac"], "inner", r.get("inner_product"))
PY
done
SMO_BENCH_TIMING_STRIDE=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench_stride1.json 2> $O/bench_stride1.err
python3 - <<PY
import json
d=json.load(open("$O/bench_stride1.json")); r=d["roofline"]
print("stride1", "ms/step", d["ms_per_step"], "sampled", r["avg_launch_ms_sampled"], "every", r["avg_launch_ms_every_launch"])
PY
timeout -k 10 600 python tools/xp_opt_warnings.py > $O/opt_warnings.txt 2>&1; tail -30 $O/opt_warnings.txt
timeout -k 10 900 python bench.py --devices 0,0,0,0,0,0,0,0 --npts 256 --iters 50 --steps 1 --warmup 1 > $O/bench_dev8_256.json 2> $O/bench_dev8_256.err; python3 -c "
import json; d=json.load(open('$O/bench_dev8_256.json')); c=d['config']; print({k:c[k] for k in ('compute_ms_per_step_pair','exchange_ms_per_step_pair','wall_ms_per_step_pair','host_issue','transpose_pull','host_rendezvous_per_step_pair')})"
timeout -k 10 600 python bench.py --npts 128 --steps 1 --warmup 1 --no-secondary --iters 100 > $O/bench_cpu.json 2> $O/bench_cpu.err; python3 -c "
import json; d=json.load(open('$O/bench_cpu.json')); print(d['cpu_baseline']); print(d['config']['cpu_single_socket'])"
