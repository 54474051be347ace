#!/bin/bash
# disassemble the gfx950 code of one translation unit of libbh.so: tools/disasm.sh bh_force  -> tools/bin/dis/bh_force.s
root=$(cd "$(dirname "$0")/.." && pwd)
D=$root/tools/bin/dis; mkdir -p $D
tu=${1:-bh_force}
objcopy -O binary --only-section=.hip_fatbin $root/nbody-barnes-hut-cuda_amd/build/$tu.o $D/$tu.fatbin &&
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$D/$tu.fatbin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$D/$tu.co &&
/opt/rocm/lib/llvm/bin/llvm-objdump -d --mcpu=gfx950 $D/$tu.co > $D/$tu.s && echo $D/$tu.s
