#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider -k "coop or fused or range or fullsize_force" > $O/pytest_r4_9.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest_r4_9.log
for rep in 1 2; do
for T in 0 2048 2560 3072 3584 4096 5120; do
  BH_FORCE_TAIL=$T BH_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libs/study.so python bench.py --bodies 1000000 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1M 0.5 tail=$T', 'ms/step', round(d['ms_per_step'],4), 'force', round(d['stages']['avg_force_ms'],4))"
done
done
for T in 0 2048 3072 4096; do
  BH_FORCE_TAIL=$T BH_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libs/study.so python bench.py --bodies 1000000 --theta 0.3 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1M 0.3 tail=$T', 'ms/step', round(d['ms_per_step'],4), 'force', round(d['stages']['avg_force_ms'],4))"
  BH_FORCE_TAIL=$T BH_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libs/study.so python bench.py --bodies 500000 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('500k tail=$T', 'ms/step', round(d['ms_per_step'],4), 'force', round(d['stages']['avg_force_ms'],4))"
  BH_FORCE_TAIL=$T BH_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libs/study.so python bench.py --bodies 2000000 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2M tail=$T', 'ms/step', round(d['ms_per_step'],4), 'force', round(d['stages']['avg_force_ms'],4))"
done
