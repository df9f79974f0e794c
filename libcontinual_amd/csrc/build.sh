#!/bin/bash
# Build libclhip.so for gfx950 (cross-compiles without a GPU).  Output stays in-tree so it travels
# to the GPU box with the gpurun snapshot.
set -e
cd "$(dirname "$0")"
OUT=../libclhip.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value"
mkdir -p obj
pids=()
for f in api conv conv2 conv3 bn elementwise head plan gemm attn vit_ops vit_plan augment; do
  if [ ! -f obj/$f.o ] || [ $f.hip -nt obj/$f.o ] || [ common.h -nt obj/$f.o ] || [ ../../include/clhip.h -nt obj/$f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o obj/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC obj/*.o -o $OUT
echo "built $(realpath $OUT)"
