#!/bin/bash
# builds an experiment variant of the library beside the product one: tools/build_variant.sh NAME "-DFLAG ..."  ->  3dr_amd/variants/libdr3hip_NAME.so
# (git-ignored; travels to the GPU box with the snapshot; load it with DR3HIP_LIB=3dr_amd/variants/libdr3hip_NAME.so)
set -e
NAME=$1; FLAGS=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=/tmp/dr3var_$NAME
rm -rf $T && mkdir -p $T/3dr_amd $T/include
cp -r $ROOT/3dr_amd/csrc $T/3dr_amd/csrc && cp -r $ROOT/include/* $T/include/
rm -f $T/3dr_amd/csrc/*.o $T/3dr_amd/csrc/.flags.*
make -C $T/3dr_amd/csrc EXTRA="$FLAGS" -j4 > $T/build.log 2>&1 || { tail -20 $T/build.log; exit 1; }
mkdir -p $ROOT/3dr_amd/variants && cp $T/3dr_amd/libdr3hip.so $ROOT/3dr_amd/variants/libdr3hip_$NAME.so
echo built 3dr_amd/variants/libdr3hip_$NAME.so
