#!/bin/bash
# rocprofv3 kernel statistics of a Poiseuille gradient at the reference script's resolution (200 of the 1000 steps)
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_round
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pois_stats -o pois -- python3 $R/tools/prof_pois.py 384 192 200 1 > $OUT/pois_under_rocprof.json 2> $OUT/pois_under_rocprof.err
cat $OUT/pois_stats/pois_kernel_stats.csv
