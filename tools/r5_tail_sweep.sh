#!/bin/bash
# force launch time (stage call bh_force, tools/force_stage_ms.py) against the number of cooperatively walked groups at
# the end of the mixed launch (BH_FORCE_TAIL, study build), three-pair-window walk (8 waves per SIMD)
cd $GRAFT_REPO_ROOT
export BH_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libs/study.so
for n in 500000 700000 1000000 2000000; do
  for T in 2048 2389 2730 3072 3400; do
    BH_FORCE_TAIL=$T timeout -k 10 100 python tools/force_stage_ms.py $n 0.5 20 2>/dev/null | sed "s/^/T=$T /" | cut -c1-60
  done
done
