#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..."  -- experimental build of the fused first-PointNet kernel next to the product
# library: so-net_amd/lib/variants/libsonet_hip_NAME.so (all other objects are the product's).  tools/fused_variants.py
# times the variants against each other and checks that their outputs are bit-identical.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
CS=$ROOT/so-net_amd/csrc; BD=$ROOT/so-net_amd/build; OUT=$ROOT/so-net_amd/lib/variants
make -C $CS >/dev/null
mkdir -p $OUT $BD/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -munsafe-fp-atomics -Wno-unused-function"
SRC=${SRC:-$CS/pointresnet_fused.hip}                       # SRC=...: another source file of the same entry points
/opt/rocm/bin/hipcc $FLAGS -I$CS "$@" -c $SRC -o $BD/variants/fused_$NAME.o
OBJS=$(ls $BD/*.o | grep -v pointresnet_fused.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $BD/variants/fused_$NAME.o -o $OUT/libsonet_hip_$NAME.so
echo built $OUT/libsonet_hip_$NAME.so
