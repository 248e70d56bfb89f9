#!/bin/bash
# round 4, first GPU call: CPU topology of the box, the new 8-way tests, the timing pre-marker A/B at 128^3
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_first
mkdir -p $O
{ nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | head -25; python3 -c "import os; print(len(os.sched_getaffinity(0)))"; free -g | head -2; } > $O/cpu.txt 2>&1
cd $R
timeout -k 10 900 python -m pytest tests/test_kdyn_multi_device_gpu.py tests/test_kdyn_slab_gpu.py -m gpu -x -q -k "8_way or null_and_short or pull_impl or errors" --durations=12 > $O/pytest_8way.log 2>&1
echo "pytest 8way rc $?"; tail -25 $O/pytest_8way.log
for pm in 1 0; do
  SMO_TIMING_PRE_MARKER=$pm timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench_pre$pm.json 2> $O/bench_pre$pm.err || { echo "bench pre$pm failed"; tail -5 $O/bench_pre$pm.err; }
  python3 - <<PY
import json
d=json.load(open("$O/bench_pre$pm.json")); r=d["roofline"]
print("pre_marker=$pm", "ms/step", d["ms_per_step"], "sampled", r["avg_launch_ms_sampled"], "every", r["avg_launch_ms_every_launch"], "frac", r["frac"], "inner", r.get("inner_product"))
PY
done
