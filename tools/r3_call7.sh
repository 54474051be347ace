#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/r3_check.sh c7 2>&1 | tee gpurun_out/check_c7.txt
bash tools/r3_dd1.sh dd3 2>&1 | tee gpurun_out/check_dd3.txt | grep -v "^void\|^(anon\|^__amd"
grep "total per rank-step\|world-1" gpurun_out/check_dd3.txt
