#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_r4_4.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -15 $O/pytest_r4_4.log
export BH_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libs/study.so
for n in 1000000 500000 300000 2000000; do
  for T in 0 1024 2048 3072 4096 6144 100000; do
    BH_FORCE_TAIL=$T python bench.py --bodies $n --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('n=$n tail=$T', 'ms/step', round(d['ms_per_step'],4), 'force', round(d['stages']['avg_force_ms'],4))"
  done
done
BH_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libs/ftrace.so timeout -k 10 300 python tools/force_trace.py 1000000 0.5 12 > $O/force_trace_mixed_1000000.txt 2>&1; head -30 $O/force_trace_mixed_1000000.txt
