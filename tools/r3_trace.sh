#!/bin/bash
# design study: block timelines of the tree build / local sort at 1M (trace build) + one step's kernel timeline
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
N=${1:-1000000}
BH_LIB_PATH=$R/tools/bin/libs/trace.so timeout -k 5 120 python3 tools/tree_trace.py $N > $O/tree_trace_$N.txt 2>&1
BH_LIB_PATH=$R/tools/bin/libs/trace.so timeout -k 5 120 python3 tools/lsort_trace.py $N > $O/lsort_trace_$N.txt 2>&1
rm -rf $O/kt_tl; rocprofv3 --kernel-trace --output-format csv -d $O/kt_tl -- python3 bench.py --steps 40 --warmup 3 --no-cpu-baseline --n $N > $O/kt_tl_bench.json 2> $O/kt_tl_err.txt
python3 tools/step_timeline.py $(ls $O/kt_tl/*/*kernel_trace.csv | head -1) 30 > $O/step_timeline_$N.txt 2>&1
cat $O/tree_trace_$N.txt $O/lsort_trace_$N.txt $O/step_timeline_$N.txt
