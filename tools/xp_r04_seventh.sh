#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
echo "== 128 dyn=1 stagger"; SMO_KD_X_DYN=1 bash tools/sweep_env.sh SMO_KD_X_DYN_STAGGER "0 4 8 16 32" 128 200
echo "== 256 dyn=1 stagger"; SMO_KD_X_DYN=1 bash tools/sweep_env.sh SMO_KD_X_DYN_STAGGER "0 8 16 32" 256 16
