#!/bin/bash
# same-box sweep of the run length of xcd_mode 2 (variants tools/bin/libs/run<R>.so built with -DBH_XCD_RUN=<R>)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in base "$@"; do
  if [ "$v" = base ]; then unset BH_LIB_PATH; else export BH_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libs/$v.so; fi
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --xcd-mode 2 $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), round(d['stages']['avg_force_ms'],4))"
done; done
