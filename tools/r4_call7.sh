#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
BH_COOP_SUBSH=13 BH_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libs/study.so timeout -k 10 600 python tools/coop_lists.py > $O/coop_lists.txt 2>&1; cat $O/coop_lists.txt
for cfg in "1000000 0.3" "1000000 0.5"; do
  set -- $cfg
  for S in 11 12 13; do
  BH_COOP_SUBSH=$S BH_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libs/study.so python bench.py --bodies $1 --theta $2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 subsh=$S', 'ms/step', round(d['ms_per_step'],4), 'force', round(d['stages']['avg_force_ms'],4))"
  done
  BH_FORCE_TAIL=0 BH_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libs/study.so python bench.py --bodies $1 --theta $2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 all one-wave', 'ms/step', round(d['ms_per_step'],4), 'force', round(d['stages']['avg_force_ms'],4))"
done
