#!/bin/bash
# LDS / memory-unit / occupancy counters of the KDyn kernels (separate --pmc passes of a short 128^3 run) -> gpurun_out/prof_stalls/summary.txt
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_stalls
N=${1:-128}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "LDSBankConflict LdsUtil" "MemUnitStalled VALUBusy" "MeanOccupancyPerActiveCU" "VmemLatency" "LdsLatency"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 $R/tools/prof_kdyn.py $N 4 > $OUT/p$i.log 2>&1 || { echo "pass $i ($set) failed"; tail -3 $OUT/p$i.log; }
done
cd $R
python3 tools/summarize_pmc.py $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5 > $OUT/summary.txt
grep -A8 "kd_x_pass<192, [23]\|kd_y_pass<192, true\|kd_z_forward<192, 1" $OUT/summary.txt | head -80
