#!/bin/bash
# tools/build_variant.sh <name> <file.hip> [-DFLAG ...] — a second libpfx build with one kernel file compiled under extra flags: gpurun_out-free A/B material
# (paintfe_amd/libpfx_<name>.so, picked up through PFX_LIB_PATH; tools/ab_libs.sh alternates two builds on one box)
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
cd $ROOT/paintfe_amd/csrc
NAME=$1; FILE=$2; shift 2
STEM=${FILE%.hip}
mkdir -p build/var_$NAME
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-gpu-flush-denormals-to-zero -Wall -Wno-unused-function --offload-arch=gfx950"
EXTRA=$(make -pn 2>/dev/null | sed -n "s/^FLAGS_$STEM := //p")
/opt/rocm/bin/hipcc $FLAGS $EXTRA "$@" -c $FILE -o build/var_$NAME/$STEM.o
OBJS=$(ls build/*.o | grep -v "build/$STEM.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpfx_$NAME.so $OBJS build/var_$NAME/$STEM.o -lz -lpthread
echo built paintfe_amd/libpfx_$NAME.so
