#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_fourth
mkdir -p $O
cd $R
timeout -k 10 300 python tools/xp_clock_probe.py 128 1000 4 > $O/clock_probe.txt 2>&1; cat $O/clock_probe.txt | grep -v amdgpu.ids
timeout -k 10 600 python tools/stress_multi_device.py 64 20 6 8 > $O/stress_w8.txt 2>&1; tail -3 $O/stress_w8.txt
timeout -k 10 600 python tools/stress_multi_device.py 48 20 6 4 > $O/stress_w4.txt 2>&1; tail -3 $O/stress_w4.txt
for ch in 1 0; do
SMO_PEER_CHAINED=$ch timeout -k 10 900 python bench.py --devices 0,0,0,0,0,0,0,0 --npts 256 --iters 50 --steps 1 --warmup 1 > $O/bench_dev8_256_ch$ch.json 2> $O/bench_dev8_256_ch$ch.err; python3 -c "
import json; d=json.load(open('$O/bench_dev8_256_ch$ch.json')); c=d['config']; print('chained=$ch', {k:c[k] for k in ('compute_ms_per_step_pair','exchange_ms_per_step_pair','wall_ms_per_step_pair','host_issue_ms_per_step_pair','host_bound_loop')})"
done
# SQ LDS counters of the z kernels: base vs the swizzled tile
cd /tmp && export TMPDIR=/tmp
for v in base zswz1; do
  if [ $v = base ]; then unset SMO_LIB; else export SMO_LIB=$R/xp_tmp/lib/libsmo_$v.so; fi
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL --kernel-trace --output-format csv -d $O/sq_$v -o p -- python3 $R/tools/prof_kdyn.py 128 4 > $O/sq_$v.log 2>&1 || echo "sq $v failed"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq2_$v -o p -- python3 $R/tools/prof_kdyn.py 128 4 > $O/sq2_$v.log 2>&1 || echo "sq2 $v failed"
done
unset SMO_LIB
cd $R
for v in base zswz1; do python3 tools/summarize_pmc.py $O/sq_$v $O/sq2_$v > $O/sq_${v}_summary.txt; done
grep -A12 "kd_z_forward" $O/sq_base_summary.txt | head -60
echo ======; grep -A12 "kd_z_forward" $O/sq_zswz1_summary.txt | head -60
timeout -k 10 900 python -m pytest tests/test_bench_gpu.py tests/test_kdyn_multi_device_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
