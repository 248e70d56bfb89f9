#!/bin/bash
# Out-of-line device calls in the SHB23 any-N kernels (csrc/shb23.hip `DctWork<0>`).
#   tools/run_outline_abi.sh build   (CPU box)  micro test + two experimental libsmo builds whose dct2<0>/dct3<0> are real calls:
#        tools/bin/libsmo_outline_A.so  the round-2 failing state: no inline attributes on dct2<0>/dct3<0>/load_tables<0>/the kernels' lambdas
#        tools/bin/libsmo_outline_B.so  today's sources with dct3<0> alone marked noinline
#   tools/run_outline_abi.sh run     (GPU box)  micro modes 1..5, then tools/diag_outline_shb.py on the product build, A and B;
#        every step under its own timeout, the chain stops at the first step that is killed; output under gpurun_out/outline/
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
CSRC=$ROOT/spheremanopt_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
if [ "${1:-}" = build ]; then
    mkdir -p tools/bin
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -save-temps=obj -o tools/bin/micro_outline_abi tools/micro_outline_abi.hip || exit 1
    n=$(grep -c s_swappc_b64 tools/bin/micro_outline_abi-hip-amdgcn-amd-amdhsa-gfx950.s)
    [ "$n" -ge 4 ] || { echo "micro test: the callees were inlined ($n calls)"; exit 1; }
    make -C "$CSRC" -j4 >/dev/null || exit 1
    W=$(mktemp -d)
    NOINL3='s/template <> __device__ __forceinline__ void dct3<0>/template <> __device__ __attribute__((noinline)) void dct3<0>/'
    NOINL2='s/template <> __device__ __forceinline__ void dct2<0>/template <> __device__ __attribute__((noinline)) void dct2<0>/'
    # name | sed script on csrc/shb23.hip | extra compiler flags
    variants=(
      "A|s/template <> __device__ __forceinline__ void dct2<0>/template <> __device__ void dct2<0>/;s/template <> __device__ __forceinline__ void dct3<0>/template <> __device__ void dct3<0>/;s/template <> __device__ __forceinline__ void load_tables<0>/template <> __device__ void load_tables<0>/;s/ __attribute__((always_inline)) {/ {/|"
      "B|$NOINL3|"
      "C|$NOINL3|-mllvm -enable-ipra=0"
      "D|$NOINL2|"
      "E|$NOINL3|-mllvm -amdgpu-waitcnt-forcezero=1"
      "F|$NOINL3|-fno-strict-aliasing"
      "G|$NOINL3|-O1"
    )
    [ -n "${VARIANTS:-}" ] || VARIANTS="A B C D E F G"
    for spec in "${variants[@]}"; do
        v=${spec%%|*}; rest=${spec#*|}; sedx=${rest%%|*}; extra=${rest#*|}
        case " $VARIANTS " in *" $v "*) ;; *) continue;; esac
        sed -e 's#"fft_lds.hpp"#"'"$CSRC"'/fft_lds.hpp"#' -e "$sedx" "$CSRC/shb23.hip" > "$W/shb23_$v.hip"
        /opt/rocm/bin/hipcc $FLAGS $extra -I"$CSRC" -save-temps=obj -c "$W/shb23_$v.hip" -o "$W/shb23_$v.o" || exit 1
        echo "variant $v ($extra): $(grep -c s_swappc_b64 "$W/shb23_$v-hip-amdgcn-amd-amdhsa-gfx950.s") out-of-line calls"
        [ -n "${KEEP_ISA:-}" ] && cp "$W/shb23_$v-hip-amdgcn-amd-amdhsa-gfx950.s" "$KEEP_ISA/shb23_$v.s"
        objs=$(ls "$CSRC"/build/*.o | grep -v shb23)
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "tools/bin/libsmo_outline_$v.so" $objs "$W/shb23_$v.o" -ldl || exit 1
    done
    rm -rf "$W"
    exit 0
fi
OUT=gpurun_out/outline
mkdir -p $OUT
ok=1
for m in ${MICRO-1 2 3 4 5}; do
    timeout -k 10 60 tools/bin/micro_outline_abi $m > $OUT/micro_$m.txt 2>&1
    rc=$?
    echo "micro mode $m: rc $rc: $(tail -1 $OUT/micro_$m.txt)"
    if [ $rc -ge 124 ]; then ok=0; break; fi      # killed: no further GPU step
done
if [ $ok = 1 ]; then
    for v in product ${VARIANTS:-A B C D E F G}; do
        lib=tools/bin/libsmo_outline_$v.so
        [ $v = product ] && lib=spheremanopt_amd/lib/libsmo.so
        [ -f $lib ] || continue
        SMO_LIB=$ROOT/$lib timeout -k 10 240 python tools/diag_outline_shb.py > $OUT/diag_$v.txt 2>&1
        rc=$?
        echo "== diag $v: rc $rc"; cat $OUT/diag_$v.txt | tail -12
        if [ $rc -ge 124 ]; then break; fi
    done
fi
exit 0
