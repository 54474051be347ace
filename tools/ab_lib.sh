#!/bin/bash
# design study: run tools/build_only.py under rocprofv3 with alternative builds of libbh.so (tools/bin/libs/<v>.so,
# selected with BH_LIB_PATH; the product library is never overwritten)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "$@"; do
  export BH_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libs/$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ab_$v -- python3 tools/build_only.py 1000000 > gpurun_out/ab_$v.log 2>&1
  grep -h "pairs_kernel\|emit_kernel" gpurun_out/prof_ab_$v/*/*_kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,100-
done
