#!/bin/bash
# Builds tools/variants/<name>/libnerf_hip.so: the product library with ONE translation unit recompiled with extra
# flags (kernel A/B experiments, tools/ab.sh).  usage: bash tools/build_variant.sh <name> <file.hip> "<extra flags>"
set -e
name=$1; src=$2; extra=$3
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/tools/variants/$name; mkdir -p $out
make -C $root/nerf-loc_amd/csrc -j8 > /dev/null
flags="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -Wno-unused-function -Wno-pass-failed"
case $src in mlp.hip|train.hip|train_bf16.hip) flags="$flags -mllvm -amdgpu-mfma-vgpr-form=1";; mlp_split.hip|mlp_split_x2.hip|mlp_split_train.hip|dgrad_split.hip) flags="$flags -mllvm -pragma-unroll-threshold=4000000";; esac
/opt/rocm/bin/hipcc $flags $extra -c $root/nerf-loc_amd/csrc/$src -o $out/${src%.hip}.o
objs=""; for o in $root/build/obj/*.o; do b=$(basename $o); if [ $b = ${src%.hip}.o ]; then objs="$objs $out/$b"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libnerf_hip.so $objs
echo built $out/libnerf_hip.so
