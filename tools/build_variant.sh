#!/bin/bash
# A variant build of libfastdiff_hip.so with probe macros on some stage files:
#   tools/build_variant.sh out.so fd_kernels_kp.hip "-DFD_GX_STORE_AUX=0"
#   tools/build_variant.sh out.so fd_kernels_lvc.hip,fd_kernels_first_final.hip "-DFD_LVC_NT=82"
# (links the variant objects with the other objects of the regular build: run `python -m fastdiff_amd.build` first)
set -eu
OUT=$1; SRCS=$2; DEFS=$3
B=fastdiff_amd/build
TMP=$(mktemp -d)
for SRC in ${SRCS//,/ }; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fno-honor-nans -x hip $DEFS -c fastdiff_amd/csrc/$SRC -o $TMP/$(basename $SRC .hip).o &
done
wait
OBJS=""
for o in $B/*.o; do [ -f $TMP/$(basename $o) ] && OBJS="$OBJS $TMP/$(basename $o)" || OBJS="$OBJS $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS
rm -rf $TMP
