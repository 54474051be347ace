#!/bin/bash
# What the own pass hides, timed on a GPU of its own: the decomposed step at world size 1 through RCCL with X4 made to
# last BH_DD_FAKE_X4_US longer (study build: one idle wave on the main stream), one pass against the first 20 / 30 %
# in two passes.  Every piece is the rank's own at world size 1: the own pass walks the whole tree for its bodies, the
# remote pass only the top record — the overlap is real, the price of two passes is not represented.
cd $GRAFT_REPO_ROOT; O=gpurun_out
export LD_PRELOAD=$GRAFT_REPO_ROOT/tools/bin/libs/study.so
for us in 0 125 250; do
  for form in "--one-pass" "--split --split-pct 20" "--split --split-pct 30"; do
    export BH_DD_FAKE_X4_US=$us
    r=$(./nbody-barnes-hut-cuda_amd/bh_bench --n 1000000 --ic plummer --steps 200 --warmup 20 --gpus 1 --dist $form --quiet 2>&1 | grep "without a sync" | sed 's/.*per frame: //')
    echo "X4 + $us us  $form : $r"
  done
done
