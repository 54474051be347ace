#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -p no:cacheprovider -k "coop" > $O/pytest_r4_2.log 2>&1
rc=$?; echo "pytest coop rc=$rc"; tail -15 $O/pytest_r4_2.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 900 python tools/coop_sweep.py > $O/coop_sweep_1.txt 2>&1; echo "sweep rc=$?"; cat $O/coop_sweep_1.txt
