#!/bin/bash
# A/B of alternative builds of libbh.so on one box: bench force time, interleaved repetitions.
# Variants live under tools/bin/libs/<name>.so and are selected with BH_LIB_PATH (the product library is
# never overwritten); "base" = the product build.   tools/force_ab.sh base sparse_T8 ...
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in "$@"; do
  if [ "$v" = base ]; then unset BH_LIB_PATH; else export BH_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libs/$v.so; fi
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), round(d['stages']['avg_force_ms'],4))"
done; done
