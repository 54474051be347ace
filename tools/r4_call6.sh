#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_r4_6.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -8 $O/pytest_r4_6.log
[ $rc -ne 0 ] && exit 1
for rep in 1 2; do
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('product 1M', 'ms/step', round(d['ms_per_step'],4), 'force', round(d['stages']['avg_force_ms'],4), 'floorfrac', round(d['roofline']['issue']['frac_of_valu_floor'],3), d['roofline']['issue']['counted_this_run'])"
  for T in 0 2048 3072 4096; do
    BH_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libs/study.so BH_FORCE_TAIL=$T python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1M tail=$T', 'ms/step', round(d['ms_per_step'],4), 'force', round(d['stages']['avg_force_ms'],4))"
  done
done
for cfg in "1000000 0.3" "500000 0.5" "125000 0.5" "65536 0.5" "16384 0.5" "8000000 0.5"; do
  set -- $cfg
  python bench.py --bodies $1 --theta $2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2', 'ms/step', round(d['ms_per_step'],4), 'force', round(d['stages']['avg_force_ms'],4))"
done
BH_FORCE_TAIL=0 BH_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libs/ftrace.so timeout -k 10 300 python tools/force_trace.py 1000000 0.5 12 > $O/force_trace_7w_1000000.txt 2>&1; head -8 $O/force_trace_7w_1000000.txt; grep resident $O/force_trace_7w_1000000.txt
python tools/coop_sweep.py 16384 65536 125000 > $O/coop_sweep_3.txt 2>&1; cat $O/coop_sweep_3.txt
