#!/bin/bash
# A variant build of libfastdiff_hip.so with probe macros on some stage files:
#   tools/build_variant.sh out.so fd_kernels_kp.hip "-DFD_GX_STORE_AUX=0"
#   tools/build_variant.sh out.so fd_kernels_lvc.hip,fd_kernels_first_final.hip "-DFD_LVC_NT=82"
# (links the variant objects with the other objects of the regular build: run `python -m fastdiff_amd.build` first)
set -eu
OUT=$1; SRCS=$2; DEFS=$3
B=fastdiff_amd/build
TMP=$(mktemp -d)
# the regular build's own compiler and flags (fastdiff_amd/build.py), so a variant differs from it by the probe macros only
HIPCC=$(python -c "from fastdiff_amd import build; print(build.HIPCC)")
FLAGS=$(python -c "from fastdiff_amd import build; print(' '.join(build.FLAGS))")
for SRC in ${SRCS//,/ }; do
  $HIPCC $FLAGS $DEFS -c fastdiff_amd/csrc/$SRC -o $TMP/$(basename $SRC .hip).o &
done
wait
OBJS=""
for o in $B/*.o; do [ -f $TMP/$(basename $o) ] && OBJS="$OBJS $TMP/$(basename $o)" || OBJS="$OBJS $o"; done
$HIPCC $(python -c "from fastdiff_amd import build; print(build.FLAGS[0])") -shared -fPIC -o $OUT $OBJS
rm -rf $TMP
