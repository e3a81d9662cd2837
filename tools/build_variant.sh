#!/bin/bash
# dev (no GPU needed): blender-ngp_amd/lib_ab/<name>/libngp_hip.so = the kernel library with network.hip compiled with extra -D flags (the other objects are the
# product build's).  usage: tools/build_variant.sh <name> "<flags>"   — then tools/gpu_ab_set.sh <name> ... on the GPU box
set -e
name=$1; flags=$2
cd "$(dirname "$0")/../blender-ngp_amd"
python build.py --kernels > /dev/null
mkdir -p lib_ab/$name build_ab/$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-variable $flags -c csrc/network.hip -o build_ab/$name/network.o
objs=""; for s in density_grid train_samples loss render multi_render comm probe; do objs="$objs build/$s.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build_ab/$name/network.o -o lib_ab/$name/libngp_hip.so -ldl
echo lib_ab/$name/libngp_hip.so
