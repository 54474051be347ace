#!/bin/bash
# SQ / SQC counter passes on the force kernel (GPU box, via gpurun).  One rocprofv3 run per group of 8
# counters (separate --pmc passes, no trace domains beside --kernel-trace).  Results: gpurun_out/sq_<tag>/pass*/
#   tools/sq_force.sh <tag> [bench.py args...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-sq}; shift
OUT=$R/gpurun_out/sq_$TAG; mkdir -p $OUT; cd $R
ARGS="--steps 3 --warmup 1 --no-cpu-baseline $*"
P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
P2="SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_DCACHE_BUSY_CYCLES SQC_TC_STALL SQC_TC_DATA_READ_REQ SQ_INST_CYCLES_SMEM"
P3="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY"
P4="GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_SALU SQ_INSTS_VSKIPPED SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES SQ_CYCLES"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pass$i -- python3 bench.py $ARGS > $OUT/pass$i.json 2> $OUT/pass$i.err || { tail -3 $OUT/pass$i.err; }
done
python3 tools/sq_summary.py $OUT
