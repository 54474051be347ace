#!/bin/bash
# kernel timeline of one steady step of the decomposed step at world size 1 through RCCL (bh_bench --gpus 1 --dist):
# the gaps at the two points where the host looks at device results (body count after the migration; X4 headers)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
rm -rf $O/prof_w1
rocprofv3 --kernel-trace --output-format csv -d $O/prof_w1 -- ./nbody-barnes-hut-cuda_amd/bh_bench --n 1000000 --ic plummer --steps 30 --warmup 10 --gpus 1 --dist --quiet > $O/w1.txt 2>&1
python3 tools/step_timeline.py $(find $O/prof_w1 -name "*kernel_trace.csv" | head -1) 55 > $O/step_timeline_world1_rccl.txt
cat $O/step_timeline_world1_rccl.txt
