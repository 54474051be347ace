#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for n in 1000000 65536; do
rm -rf $O/prof_tl_$n
rocprofv3 --kernel-trace --output-format csv -d $O/prof_tl_$n -- python3 bench.py --bodies $n --steps 40 --warmup 5 --no-cpu-baseline > $O/prof_tl_$n.json 2> $O/prof_tl_$n.err
f=$(find $O/prof_tl_$n -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f 30 > $O/step_timeline_$n.txt; cat $O/step_timeline_$n.txt
done
