#!/bin/bash
# A/B of alternative builds (tools/bin/libs/<v>.so via BH_LIB_PATH; "base" = the product library): sort time
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = base ]; then unset BH_LIB_PATH; else export BH_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libs/$v.so; fi
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), 'sort', round(d['stages']['last_step_ms']['sort'],4))"
done; done
