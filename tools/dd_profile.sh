#!/bin/bash
# GPU box: kernel trace of the in-process 8-rank rehearsal of the domain-decomposed step
# (8 x 1M bodies on one GPU; per-rank kernel time = per-kernel total / (steps x ranks)).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
W=${1:-8}; N=${2:-8000000}; S=${3:-6}; TAG=${4:-dd}
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 tools/dd_debug.py --world $W --n $N --steps $S $DDFLAGS > $OUT/prof_$TAG.out 2> $OUT/prof_$TAG.err
find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -3
