#!/bin/bash
# A variant build of libfastdiff_hip.so with probe macros on ONE stage file:  tools/build_variant.sh out.so fd_kernels_kp.hip "-DFD_GX_STORE_AUX=2"
# (links the variant object with the other objects of the regular build: run `python -m fastdiff_amd.build` first)
set -eu
OUT=$1; SRC=$2; DEFS=$3
B=fastdiff_amd/build
O=/tmp/variant_$$.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fno-honor-nans -x hip $DEFS -c fastdiff_amd/csrc/$SRC -o $O
OBJS=""
for o in $B/*.o; do [ "$(basename $o .o)" = "$(basename $SRC .hip)" ] && OBJS="$OBJS $O" || OBJS="$OBJS $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS
rm -f $O
