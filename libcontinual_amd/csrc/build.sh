#!/bin/bash
# Build libclhip.so for gfx950 (cross-compiles without a GPU).  Output stays in-tree so it travels
# to the GPU box with the gpurun snapshot.
set -e
cd "$(dirname "$0")"
OUT=../libclhip.so
OBJ=obj
# ABL=1: ablation / phase-trace build of the conv4 kernels into libclhip_abl.so (tools/ubench/conv_bench_abl)
if [ -n "$ABL" ]; then OUT=../libclhip_abl.so; OBJ=obj_abl; EXTRA="-DCLHIP_ABLATION"; fi
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value $EXTRA"
mkdir -p $OBJ
pids=()
for f in api conv conv2 conv3 conv4 conv5 conv6 conv7 conv8 conv9 stage stage_train wgrad4 stem shortcut bn elementwise head plan gemm gemm8 attn vit_ops vit_plan augment; do
  if [ ! -f $OBJ/$f.o ] || [ $f.hip -nt $OBJ/$f.o ] || [ common.h -nt $OBJ/$f.o ] || [ xch.h -nt $OBJ/$f.o ] || [ ../../include/clhip.h -nt $OBJ/$f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o $OBJ/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC $OBJ/*.o -o $OUT
echo "built $(realpath $OUT)"
