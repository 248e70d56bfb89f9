#!/bin/bash
# Copy what tools/profile_round.sh left under gpurun_out/prof_round/ into profiles/ under this round's names.
# usage: tools/copy_round_profiles.sh r04
set -e
cd "$(dirname "$0")/.."
R=${1:?round prefix, e.g. r04}
O=gpurun_out/prof_round
cp $O/pmc_128.json profiles/${R}_kdyn128_pmc.json
cp $O/pmc_128_summary.txt profiles/${R}_kdyn128_pmc_summary.txt
cp $O/pmc_256.json profiles/${R}_kdyn256_pmc.json
cp $O/pmc_256_summary.txt profiles/${R}_kdyn256_pmc_summary.txt
cp $O/bench_stats/kdyn128_kernel_stats.csv profiles/${R}_kdyn128_bench_kernel_stats.csv
cp $O/bench_under_rocprof.json profiles/${R}_kdyn128_bench_under_rocprof.json
cp $O/bench_stats_256/kdyn256_kernel_stats.csv profiles/${R}_kdyn256_bench_kernel_stats.csv
cp $O/bench256_under_rocprof.json profiles/${R}_kdyn256_bench_under_rocprof.json
cp $O/pois_stats/pois_kernel_stats.csv profiles/${R}_poiseuille_384x192_kernel_stats.csv
cp $O/pmc_pois_summary.txt profiles/${R}_poiseuille_384x192_pmc_summary.txt
cp $O/pois_under_rocprof.json profiles/${R}_poiseuille_384x192_iters200_under_rocprof.json
grep -h source_sha profiles/${R}_kdyn128_pmc.json profiles/${R}_kdyn256_pmc.json
