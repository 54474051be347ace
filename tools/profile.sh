#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + HBM PMC passes of the bench command.
# Outputs land in gpurun_out/prof_*; summaries are copied into profiles/ by hand afterwards.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/prof_trace.json 2> $OUT/prof_trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_fetch -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/prof_fetch.json 2> $OUT/prof_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_write -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/prof_write.json 2> $OUT/prof_write.err
find $OUT -name "*.csv" | head -30
