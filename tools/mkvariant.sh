#!/bin/bash
# Build an alternative libbh.so with extra compiler flags into tools/bin/libs/<name>.so (never the product
# library); select it at run time with BH_LIB_PATH.   tools/mkvariant.sh look32 -DBH_OS_LOOK=32
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
# SRC_COMMIT=<git rev>: build that revision's sources instead of the working tree (A/B against an older state)
src=$root
if [ -n "$SRC_COMMIT" ]; then
  src=$root/tools/bin/src_$name; rm -rf $src; mkdir -p $src
  git -C $root archive $SRC_COMMIT include nbody-barnes-hut-cuda_amd/csrc | tar -x -C $src
fi
pkg=$src/nbody-barnes-hut-cuda_amd
out=$root/tools/bin/libs; bld=$root/tools/bin/build_$name
mkdir -p $out $bld
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-inline-asm -fno-slp-vectorize -I$src/include -I$pkg/csrc"
pids=()
for f in bh_api bh_scan bh_sort bh_sort_onesweep bh_tree bh_force bh_dd bh_group; do
  /opt/rocm/bin/hipcc $flags "$@" -c $pkg/csrc/$f.hip -o $bld/$f.o 2> $bld/$f.log & pids+=($!)
done
for f in bh_ic bh_io; do
  /opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -ffp-contract=off -I$src/include -I$pkg/csrc -c $pkg/csrc/$f.cpp -o $bld/$f.o 2> $bld/$f.log & pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/$name.so $bld/*.o -ldl -lpthread
echo built $out/$name.so
