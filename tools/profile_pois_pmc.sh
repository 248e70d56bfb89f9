#!/bin/bash
# HBM traffic counters (separate --pmc passes) of a short Poiseuille gradient at the reference script's resolution
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_round
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pois_pmc_fetch -o f -- python3 $R/tools/prof_pois.py 384 192 20 1 > $OUT/pois_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pois_pmc_write -o w -- python3 $R/tools/prof_pois.py 384 192 20 1 > $OUT/pois_pmc_write.log 2>&1
cd $R
python3 tools/summarize_pmc.py $OUT/pois_pmc_fetch $OUT/pois_pmc_write
