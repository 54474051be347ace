#!/bin/bash
# GPU box: VALU/SALU instruction counts of the force kernel, plain engine vs domain-decomposed ranks
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/sq_plain1 -- python3 bench.py --bodies 1000000 --seed 3 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/sq_plain1.json 2> $OUT/sq.err
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/sq_plain8 -- python3 bench.py --bodies 8000000 --seed 3 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/sq_plain8.json 2>> $OUT/sq.err
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/sq_dd1 -- python3 tools/dd_debug.py --world 1 --n 1000000 --steps 3 > $OUT/sq_dd1.out 2>> $OUT/sq.err
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/sq_dd8 -- python3 tools/dd_debug.py --world 8 --n 8000000 --steps 3 > $OUT/sq_dd8.out 2>> $OUT/sq.err
tail -3 $OUT/sq.err
