#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_dd.py tests/test_gpu_dist.py -m gpu -q -x -s -p no:cacheprovider > $O/pytest_r4_11.log 2>&1
rc=$?; echo "pytest dd rc=$rc"; grep -E "boundaries moved|8 ranks x 1M|passed|failed|Error" $O/pytest_r4_11.log | tail -8
[ $rc -ne 0 ] && tail -30 $O/pytest_r4_11.log
timeout -k 10 600 python tools/dd_debug.py --world 8 --n 8000000 --steps 12 --no-split > $O/dd_step_log_8x1M.txt 2>&1; grep "^step" $O/dd_step_log_8x1M.txt | cut -c1-420
