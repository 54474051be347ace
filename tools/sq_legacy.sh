#!/bin/bash
# SQ counter passes on the bench  (GPU box). Results: gpurun_out/prof_sq_v*
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
for v in 0; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/prof_sq_v$v -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_sq_v$v.json 2> $OUT/prof_sq.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/prof_sq2_v$v -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_sq2_v$v.json 2>> $OUT/prof_sq.err
done
tail -2 $OUT/prof_sq.err
