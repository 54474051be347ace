#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_r4_8.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -8 $O/pytest_r4_8.log
[ $rc -ne 0 ] && exit 1
for cfg in "1000000 0.3" "1000000 0.5" "500000 0.5" "2000000 0.5" "1000000 0.7"; do
  set -- $cfg
  python bench.py --bodies $1 --theta $2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 product', 'ms/step', round(d['ms_per_step'],4), 'force', round(d['stages']['avg_force_ms'],4))"
  BH_FORCE_TAIL=0 BH_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libs/study.so python bench.py --bodies $1 --theta $2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 all one-wave', 'ms/step', round(d['ms_per_step'],4), 'force', round(d['stages']['avg_force_ms'],4))"
done
for T in 2048 3072 4096 6144; do
  BH_FORCE_TAIL=$T BH_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libs/study.so python bench.py --bodies 1000000 --theta 0.3 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1M 0.3 tail=$T', 'ms/step', round(d['ms_per_step'],4), 'force', round(d['stages']['avg_force_ms'],4))"
done
./nbody-barnes-hut-cuda_amd/bh_bench --n 500000 --steps 200 --warmup 20 --quiet 2>&1 | tail -3
