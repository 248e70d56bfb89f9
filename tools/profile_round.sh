#!/bin/bash
# Round profile (run on the GPU box from the repository root; writes under gpurun_out/prof_round/):
#   1. rocprofv3 --kernel-trace --stats of the benchmark command itself at 128^3 AND at 256^3 (per-kernel average durations to set against
#      bench.py's timestamps; 256^3: 40 of the 1000 steps x 3 gradients = 120 launches per kernel class, steady state — round 3's two cold
#      launches of tools/prof_kdyn.py were not)
#   2. HBM traffic counters, separate --pmc passes (FETCH_SIZE | WRITE_SIZE), of a short KDyn run at 128^3 and 256^3
#      -> pmc_<N>.json stamped with the sha of the kernel sources (bench.py quotes `traffic` only from a summary of the same sources)
#   3. kernel statistics and HBM traffic counters of the Poiseuille path
# usage: tools/profile_round.sh [bench steps]
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_round
STEPS=${1:-5}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -o kdyn128 -- python3 $R/bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-secondary --no-host-vectors > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
echo "stats 128 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats_256 -o kdyn256 -- python3 $R/bench.py --npts 256 --iters 40 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-host-vectors > $OUT/bench256_under_rocprof.json 2> $OUT/bench256_under_rocprof.err
echo "stats 256 done"
for N in 128 256; do
  IT=4; [ $N = 256 ] && IT=2
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_${N}/fetch -o f -- python3 $R/tools/prof_kdyn.py $N $IT > $OUT/pmc_${N}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_${N}/write -o w -- python3 $R/tools/prof_kdyn.py $N $IT > $OUT/pmc_${N}_write.log 2>&1
  echo "pmc $N done"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pois_stats -o pois -- python3 $R/tools/prof_pois.py 384 192 200 1 > $OUT/pois_under_rocprof.json 2> $OUT/pois_under_rocprof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_pois/fetch -o f -- python3 $R/tools/prof_pois.py 384 192 40 1 > $OUT/pmc_pois_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_pois/write -o w -- python3 $R/tools/prof_pois.py 384 192 40 1 > $OUT/pmc_pois_write.log 2>&1
echo "pois done"
cd $R
for N in 128 256; do
  python3 tools/summarize_pmc.py $OUT/pmc_${N} > $OUT/pmc_${N}_summary.txt
  python3 tools/pmc_to_json.py $OUT/pmc_${N} $OUT/pmc_${N}.json $N || true
done
python3 tools/summarize_pmc.py $OUT/pmc_pois > $OUT/pmc_pois_summary.txt
find $OUT -name "*kernel_stats.csv"
