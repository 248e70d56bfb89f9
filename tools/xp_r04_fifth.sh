#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
echo "== 256"; bash tools/sweep_variants.sh "base yh zh yzh base" 256 16
echo "== 128"; bash tools/sweep_variants.sh "base yhall base" 128 200
