#!/bin/bash
# Developer tool: A/B builds of ONE translation unit. tests/tools_variant_build.sh TAG UNIT "-DFLAG=.." -> polympc_amd/_variants/lib_TAG.so
# (the other objects come from polympc_amd/_build; select the variant with PMPC_LIB=polympc_amd/_variants/lib_TAG.so)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; UNIT=${2:-pmpc_model_robot}; shift 2 || true
mkdir -p $R/polympc_amd/_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$R/include "$@" -c -o $R/polympc_amd/_variants/${UNIT}_$TAG.o $R/polympc_amd/csrc/$UNIT.hip
OBJS=$(ls $R/polympc_amd/_build/*.o | grep -v "/$UNIT.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/polympc_amd/_variants/lib_$TAG.so $OBJS $R/polympc_amd/_variants/${UNIT}_$TAG.o
rm -f $R/polympc_amd/_variants/${UNIT}_$TAG.o
echo built $R/polympc_amd/_variants/lib_$TAG.so
