#!/bin/bash
# Experimental build of libsmo with extra compiler flags for csrc/kdyn.hip: tools/build_variant.sh NAME [flags...] -> xp_tmp/lib/libsmo_NAME.so
# (loaded through SMO_LIB by tools/sweep_variants.sh; xp_tmp/ is git-ignored scratch that still travels with gpurun)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
CS=spheremanopt_amd/csrc
mkdir -p xp_tmp/lib xp_tmp/obj_$name
make -C $CS -j4 >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -mllvm -enable-ipra=0 "$@" -c $CS/kdyn.hip -o xp_tmp/obj_$name/kdyn.hip.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o xp_tmp/lib/libsmo_$name.so $(ls $CS/build/*.o | grep -v kdyn.hip.o) xp_tmp/obj_$name/kdyn.hip.o -ldl
echo "built xp_tmp/lib/libsmo_$name.so ($*)"
