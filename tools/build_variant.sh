#!/bin/bash
# A/B build of the library with extra -D flags: tools/build_variant.sh NAME "-DPA_RT1_PD=4 ..."  -> patchaugnet_amd/csrc/ab/libpa_NAME.so
# (select it in a run with PA_LIB_PATH=patchaugnet_amd/csrc/ab/libpa_NAME.so; only the sources named in $3.. are rebuilt with the flags)
set -e
NAME=$1; FLAGS=$2; shift 2
SRCS=${@:-mlp_chain.hip}
EXP_ONLY=" fpx_reg.hip knn_lane.hip "
cd "$(dirname "$0")/../patchaugnet_amd/csrc"
mkdir -p ab/$NAME
OBJS=""
for f in *.hip; do
  [[ "$EXP_ONLY" == *" $f "* ]] && continue
  if [[ " $SRCS " == *" $f "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fno-vectorize -fvisibility=hidden -Wno-unused-function $FLAGS -c $f -o ab/$NAME/${f%.hip}.o
    OBJS="$OBJS ab/$NAME/${f%.hip}.o"
  else
    OBJS="$OBJS ${f%.hip}.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libpa_$NAME.so $OBJS
echo built ab/libpa_$NAME.so
