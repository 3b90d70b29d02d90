#!/bin/bash
# A/B arms are PRODUCT builds: tools/build_variant.sh NAME "-DVB_X=.. -DVB_Y=.." builds gemm.hip, layer.hip and attention.hip with the extra defines into
# visualbert_amd/csrc/build_NAME/ and links it with the other (unchanged) product objects -> tools/libvisualbert_hip_ab_NAME.so
# (same ABI as the in-tree library: bench.py --lib-path / tools/gpu_small_batch_ab.sh take it as the reference arm).
set -e
NAME=${1:?usage: build_variant.sh NAME "extra -D flags"}; FLAGS=$2
cd "$(dirname "$0")/../visualbert_amd/csrc"
mkdir -p build_$NAME
HF="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off"
/opt/rocm/bin/hipcc $HF $FLAGS -c gemm.hip -o build_$NAME/gemm.o &
/opt/rocm/bin/hipcc $HF $FLAGS -c layer.hip -o build_$NAME/layer.o &
/opt/rocm/bin/hipcc $HF $FLAGS -c attention.hip -o build_$NAME/attention.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libvisualbert_hip_ab_$NAME.so build_$NAME/gemm.o build_$NAME/layer.o build_$NAME/attention.o \
  build/layernorm.o build/loss.o build/optim.o build/misc.o build/heads.o build/comm.o -ldl
echo "built tools/libvisualbert_hip_ab_$NAME.so with: $FLAGS"
