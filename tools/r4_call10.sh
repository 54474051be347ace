#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_r4_10.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest_r4_10.log
for n in 16384 32768 65536 125000 160000 200000 250000 300000 500000 1000000; do
  python bench.py --bodies $n --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n product', 'ms/step', round(d['ms_per_step'],4), 'force', round(d['stages']['avg_force_ms'],4), {k: round(v,4) for k,v in d['stages']['last_step_ms'].items()})"
  if [ $n -ge 125000 ] && [ $n -le 300000 ]; then
  BH_FORCE_TAIL=100000 BH_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libs/study.so python bench.py --bodies $n --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n all-coop K=4', 'ms/step', round(d['ms_per_step'],4), 'force', round(d['stages']['avg_force_ms'],4))"
  fi
done
