#!/bin/bash
# tools/build_variant.sh TAG file.hip "-DFLAG ..." : experiment builds -- libmm2amd.so with one HIP file recompiled under extra flags,
# written to minimap2_amd/variants/libmm2amd_TAG.so (git-ignored; travels to the GPU box).  Measurement scaffolding only.
set -e
R=$(cd $(dirname $0)/.. && pwd); B=$R/minimap2_amd/build; V=$R/minimap2_amd/variants; mkdir -p $V
TAG=$1; SRC=$2; shift 2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -I$R/include "$@" -x hip -c $R/minimap2_amd/csrc/$SRC -o $V/$SRC.$TAG.o
OBJS=$(ls $B/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libmm2amd_$TAG.so $OBJS $V/$SRC.$TAG.o -lpthread
echo $V/libmm2amd_$TAG.so
