#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for N in 128 256; do
  IT=200; [ $N = 256 ] && IT=16
  echo "== $N"; bash tools/sweep_env.sh SMO_KD_X_DYN "0 1 2 3 0" $N $IT
done
echo "== wg sweep 256 adj"; SMO_KD_X_DYN=3 bash tools/sweep_env.sh SMO_KD_X_DYN_WG "2 3 4" 256 16
echo "== wg sweep 128 fwd"; SMO_KD_X_DYN=2 bash tools/sweep_env.sh SMO_KD_X_DYN_WG "3 4 5" 128 200
