#!/bin/bash
# What the main stream's kernels (LET chain, X4, validation, top trees) do while a force pass of the side stream holds
# the GPU: kernel timeline of the partial two-pass step at world size 1 through RCCL (split forced on; every piece is the
# rank's own, the remote pass walks the top only).  Study build (tools/mkvariant.sh study -DBH_STUDY):
#   at_once_prio   own pass launched at once on a lowest-priority stream (the form up to mid round 5)
#   after_let_prio own pass behind the LET export, lowest-priority stream
#   after_let_mask own pass behind the LET export, side stream masked to all but 16 CUs (the product)
#   at_once_mask32 own pass at once, side stream masked to all but 32 CUs
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
export LD_PRELOAD=$R/tools/bin/libs/study.so
run() {  # name
  rm -rf $O/prof_split_$1
  rocprofv3 --kernel-trace --output-format csv -d $O/prof_split_$1 -- ./nbody-barnes-hut-cuda_amd/bh_bench --n 1000000 --ic plummer --steps 30 --warmup 10 --gpus 1 --dist --split --split-pct 30 --quiet > $O/split_$1.txt 2>&1
  tail -2 $O/split_$1.txt
  python3 tools/split_timeline.py $(find $O/prof_split_$1 -name "*kernel_trace.csv" | head -1) 25 > $O/split_timeline_world1_$1.txt
  echo "== $1"; cat $O/split_timeline_world1_$1.txt
}
BH_DD_OWN_AT_ONCE=1 BH_DD_RESERVE_CUS=0 run at_once_prio
unset BH_DD_OWN_AT_ONCE
BH_DD_RESERVE_CUS=0 run after_let_prio
unset BH_DD_RESERVE_CUS
run after_let_mask
BH_DD_OWN_AT_ONCE=1 BH_DD_RESERVE_CUS=32 run at_once_mask32
