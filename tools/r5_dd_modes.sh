#!/bin/bash
# wall time per step of the one-GPU rehearsal (8 ranks x 1M through bh_bench --devices) for the force-pass modes
cd $GRAFT_REPO_ROOT
DEV=0,0,0,0,0,0,0,0
for rep in 1 2; do
for mode in "--one-pass" "--split --split-pct 100" "--split"; do
  echo "mode [$mode]"; timeout -k 10 200 ./nbody-barnes-hut-cuda_amd/bh_bench --n ${1:-8000000} --ic plummer --steps ${2:-12} --warmup 4 --devices $DEV --quiet $mode | tail -2 | cut -c1-330
done; done
