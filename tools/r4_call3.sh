#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "coop" > $O/pytest_r4_3.log 2>&1
rc=$?; echo "pytest coop rc=$rc"; tail -5 $O/pytest_r4_3.log
[ $rc -ne 0 ] && exit 1
echo "== touch prefetch (product build)"
timeout -k 10 900 python tools/coop_sweep.py > $O/coop_sweep_touch.txt 2>&1; echo "sweep rc=$?"; cat $O/coop_sweep_touch.txt
echo "== no touch (study build, BH_COOP_NOTOUCH=1)"
BH_COOP_NOTOUCH=1 BH_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libs/study.so timeout -k 10 900 python tools/coop_sweep.py 65536 500000 1000000 > $O/coop_sweep_notouch.txt 2>&1; cat $O/coop_sweep_notouch.txt
