#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
echo "== 128"; bash tools/sweep_env.sh SMO_KD_X_DYN "0 1" 128 200
echo "== 256"; bash tools/sweep_env.sh SMO_KD_X_DYN "0 1" 256 16
