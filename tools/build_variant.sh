#!/bin/bash
# Builds a variant of the library for interleaved A/B runs:  bash tools/build_variant.sh NAME [-DMACRO=value ...]
# -> skelsplat_amd/ab_NAME.so (git-ignored, but it travels to the GPU box); use with SKS_LIB_OVERRIDE=skelsplat_amd/ab_NAME.so
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
cd "$root/skelsplat_amd/csrc"
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function -mllvm -disable-machine-sink"
tmp=$(mktemp -d)
for s in sks_raster sks_ops sks_loop; do /opt/rocm/bin/hipcc $F "$@" -c -o $tmp/$s.o $s.hip & done
/opt/rocm/bin/hipcc $F -fno-slp-vectorize "$@" -c -o $tmp/sks_ssim.o sks_ssim.hip &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o "$root/skelsplat_amd/ab_$name.so" $tmp/*.o && echo "built skelsplat_amd/ab_$name.so"
rm -rf $tmp
