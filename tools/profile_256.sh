#!/bin/bash
# 256^3 (G = 384 kernels): kernel statistics, HBM traffic counters (separate --pmc passes, FETCH_SIZE / WRITE_SIZE) and the stall set of a
# short run -> gpurun_out/prof_256/.  Run on the GPU box from the repository root.  usage: tools/profile_256.sh [npts] [iters]
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
N=${1:-256}; IT=${2:-2}
OUT=$R/gpurun_out/prof_$N
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $R/tools/prof_kdyn.py $N $IT > $OUT/stats.log 2>&1
echo "stats done"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "LDSBankConflict LdsUtil" "MemUnitStalled VALUBusy" "MeanOccupancyPerActiveCU" "VmemLatency" "L2CacheHit"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 $R/tools/prof_kdyn.py $N $IT > $OUT/p$i.log 2>&1 || { echo "pass $i ($set) failed"; tail -3 $OUT/p$i.log; }
  echo "pass $i ($set) done"
done
cd $R
python3 tools/summarize_pmc.py $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5 $OUT/p6 $OUT/p7 > $OUT/summary.txt
python3 tools/pmc_to_json.py $OUT $OUT/pmc.json $N || true
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
head -20 $OUT/kernel_stats.csv
