#!/bin/bash
# Measurement variants of the library: tools/build_variant.sh TAG FILE.hip [-DFLAG=..]...  compiles ONE translation unit with extra
# flags, links it with the current objects of the others into hybrid-ctunet_amd/csrc/variants/libctunet_hip_TAG.so (git-ignored,
# travels to the GPU box).  A/B tools pick it up through CTU_LIB_VARIANT=TAG (tools/_variant.py) - the product never does.
set -e
cd "$(dirname "$0")/../hybrid-ctunet_amd/csrc"
tag=$1; src=$2; shift 2
make -j4 >/dev/null
mkdir -p variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c "$src" -o "variants/${src%.hip}_$tag.o"
objs=""
for f in igemm gemm_dma gemm_narrow conv3_halo norm_elementwise ff_fused pwa_fused attention attention_mfma loss_optim infer dropout comm plan; do
  if [ "$f.hip" = "$src" ]; then objs="$objs variants/${f}_$tag.o"; else objs="$objs $f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "variants/libctunet_hip_$tag.so" $objs -ldl
ls -la "variants/libctunet_hip_$tag.so"
