#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/r3_check.sh c12 2>&1 | tee gpurun_out/check_c12.txt
bash tools/r3_dd1.sh dd4 2>&1 | tee gpurun_out/check_dd4.txt | grep -v "^void\|^(anon\|^__amd"
grep "total per rank-step\|world-1" gpurun_out/check_dd4.txt
