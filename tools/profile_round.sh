#!/bin/bash
# Round profile: rocprofv3 kernel statistics of the benchmark command and HBM traffic counters (separate --pmc passes) of a short KDyn run.
# Run on the GPU box from the repository root; writes under gpurun_out/prof_round/.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_round
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -o kdyn128 -- python3 $R/bench.py --no-cpu-baseline --no-secondary > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- python3 $R/tools/prof_kdyn.py 128 4 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- python3 $R/tools/prof_kdyn.py 128 4 > $OUT/pmc_write.log 2>&1
echo "pmc done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pois_stats -o pois -- python3 $R/tools/prof_pois.py 384 192 200 1 > $OUT/pois_under_rocprof.json 2> $OUT/pois_under_rocprof.err
echo "pois done"
cd $R
python3 tools/summarize_pmc.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_summary.txt
python3 tools/pmc_to_json.py $OUT $OUT/pmc.json 128 || true
find $OUT -name "*kernel_stats.csv" | head
