#!/bin/bash
# L2 (TCC) hit/miss counters of the force kernel (GPU box, via gpurun): tools/tcc_force.sh <tag> [bench.py args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-tcc}; shift
OUT=$R/gpurun_out/tcc_$TAG; mkdir -p $OUT; cd $R
ARGS="--steps 3 --warmup 1 --no-cpu-baseline $*"
i=0
for P in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_PROBE_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pass$i -- python3 bench.py $ARGS > $OUT/pass$i.json 2> $OUT/pass$i.err || { tail -3 $OUT/pass$i.err; }
done
python3 tools/sq_summary.py $OUT
