#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for n in 1000000 8000000; do for v in 0 1; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sab_${n}_$v -- python3 tools/sort_only.py $n $v > gpurun_out/sab.log 2>&1
done; done
