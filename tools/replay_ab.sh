#!/bin/bash
# same-box A/B of library builds on the replayed force phase of an 8 x 1M rehearsal (bh_bench --replay, LD_PRELOAD):
# mean over the ranks of the one-pass form, X4 = 0.   tools/replay_ab.sh base <variant> ...
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in "$@"; do
  if [ "$v" = base ]; then unset LD_PRELOAD; else export LD_PRELOAD=$GRAFT_REPO_ROOT/tools/bin/libs/$v.so; fi
  r=$(./nbody-barnes-hut-cuda_amd/bh_bench --n 8000000 --ic plummer --devices 0,0,0,0,0,0,0,0 --steps 8 --warmup 4 --quiet --replay 2>&1 | grep "^mean | one pass")
  unset LD_PRELOAD
  echo "$v $r"
done; done
