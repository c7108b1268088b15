#!/bin/bash
# A variant library for A/B runs (K4_LIB=4k-nerf_amd/lib4k_hip_<tag>.so): the product's sources with extra -D flags on k4_march.hip only.
# usage: bash tools/build_variant.sh <tag> -DFLAG=VALUE ...      (build container; the .so travels to the GPU box with the snapshot)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; shift
P=$R/4k-nerf_amd
python $P/build.py > /dev/null
mkdir -p $P/build_$TAG
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -I$R/include -I$P/csrc "$@" -c $P/csrc/k4_march.hip -o $P/build_$TAG/k4_march.hip.o
OBJS=$(ls $P/build/*.o | grep -v k4_march.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/lib4k_hip_$TAG.so $P/build_$TAG/k4_march.hip.o $OBJS
echo $P/lib4k_hip_$TAG.so
