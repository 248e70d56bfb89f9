#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
echo "== 256"; bash tools/sweep_variants.sh "base pkB pkC pkD pkE base pkB" 256 24
