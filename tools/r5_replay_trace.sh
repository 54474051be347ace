#!/bin/bash
# kernel timelines of replayed force phases of ranks 0 and 1 of an 8 x 1M rehearsal (tools/replay_timeline.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for q in ${RANKS:-0 1}; do
rm -rf $O/prof_replay$q
rocprofv3 --kernel-trace --output-format csv -d $O/prof_replay$q -- ./nbody-barnes-hut-cuda_amd/bh_bench --n 8000000 --ic plummer --devices 0,0,0,0,0,0,0,0 --steps 6 --warmup 4 --quiet --replay-rank $q > $O/replay_rank$q.txt 2>&1
tail -6 $O/replay_rank$q.txt
f=$(find $O/prof_replay$q -name "*kernel_trace.csv" | head -1)
for cell in "0 0" "1 0" "1 1"; do python3 tools/replay_timeline.py $f $cell > $O/replay_timeline_rank${q}_$(echo $cell | tr ' ' _).txt; cat $O/replay_timeline_rank${q}_$(echo $cell | tr ' ' _).txt; done
rm -rf $O/prof_replay$q
done
